// Environment bookkeeping executed by ONE wave per graph (E4-E8 of SURVEY.md 8a).
// Operation order follows the reference exactly:
//   scatter -> (IRP load) -> cur -> done-reduction -> depot fix-ups ->
//   all-visited release -> (IRP) capacity overlay
// gym_vrp/envs/tsp.py:83-101,131-148  vrp.py:13-37  irp.py:75-99,126-155.
#pragma once
#include "common.h"

struct EnvStepOut {
  double dist;   // fp64 Euclidean edge length (vrp_graph.py:137-146)
  bool done;     // visited row all ones BEFORE the fix-ups (tsp.py:95,103-104)
};

// fp64 sqrt(dx*dx+dy*dy) with separately rounded products (no FMA contraction),
// which is what numpy evaluates for a 2-vector norm.
__device__ __forceinline__ double edge_length(const double *pos, int N, int b, int i, int j) {
  const double *p = pos + (size_t)b * N * 2;
  double dx = p[2 * i] - p[2 * j];
  double dy = p[2 * i + 1] - p[2 * j + 1];
  return sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}

// Depot fix-ups + mask write for graph b; `v0`,`v1` hold this lane's visited flags
// for nodes lane and lane+64.  at_depot = (current_location == depot).
__device__ __forceinline__ void env_fixups_and_mask(const vrp_env &e, int b, int lane,
                                                    int at_depot, int &v0, int &v1,
                                                    double load, uint8_t *mask_out,
                                                    int dep_known = -1) {
  // (a caller that already holds the depot index passes it: read here, behind the caller's
  // stores, it is a second dependent round trip)
  const int N = e.N, dep = dep_known >= 0 ? dep_known : e.depot[b];
  const int n0 = lane, n1 = lane + 64;
  // (a) disallow staying on the depot  tsp.py:141-142
  if (at_depot) { if (n0 == dep) v0 = 1; if (n1 == dep) v1 = 1; }
  // (b) VRP/IRP: the depot is open again once the vehicle has left it  vrp.py:28-31
  else if (e.kind != VRP_KIND_TSP) { if (n0 == dep) v0 = 0; if (n1 == dep) v1 = 0; }
  // (c) solved graphs may idle on the depot  tsp.py:145-146
  int ok = (n0 >= N || v0) && (n1 >= N || v1);
  if (__all(ok)) { if (n0 == dep) v0 = 0; if (n1 == dep) v1 = 0; }
  uint8_t *vis = e.visited + (size_t)b * N;
  uint8_t *mo = mask_out + (size_t)b * N;
  int m0 = v0, m1 = v1;
  if (e.kind == VRP_KIND_IRP) {  // capacity overlay on a copy  irp.py:151-153
    const double *dem = e.demand + (size_t)b * N;
    if (n0 < N && dem[n0] - load > 0.0) m0 = 1;
    if (n1 < N && dem[n1] - load > 0.0) m1 = 1;
  }
  if (n0 < N) { vis[n0] = (uint8_t)v0; mo[n0] = (uint8_t)m0; }
  if (n1 < N) { vis[n1] = (uint8_t)v1; mo[n1] = (uint8_t)m1; }
}

// One env.step for graph b with action a (wave-uniform).  All 64 lanes call it.
__device__ __forceinline__ EnvStepOut env_step_wave(const vrp_env &e, int b, int a, int lane,
                                                    uint8_t *mask_out) {
  const int N = e.N;
  const uint8_t *vis = e.visited + (size_t)b * N;
  const int n0 = lane, n1 = lane + 64;
  int v0 = (n0 < N) ? vis[n0] : 1;
  int v1 = (n1 < N) ? vis[n1] : 1;
  if (n0 == a) v0 = 1;  // tsp.py:86
  if (n1 == a) v1 = 1;
  const int src = e.cur[b];
  const int dep = e.depot[b];
  EnvStepOut out;
  out.dist = edge_length(e.pos, N, b, src, a);
  double load = 1.0;
  if (e.kind == VRP_KIND_IRP) {  // irp.py:80-86
    load = e.load[b] - e.demand[(size_t)b * N + a];
    if (a == dep) load = 1.0;
  }
  out.done = __all(v0 && v1);  // evaluated before generate_mask (tsp.py:95)
  env_fixups_and_mask(e, b, lane, a == dep, v0, v1, load, mask_out);
  if (lane == 0) {
    e.cur[b] = a;  // tsp.py:90
    if (e.kind == VRP_KIND_IRP) e.load[b] = load;
  }
  return out;
}
