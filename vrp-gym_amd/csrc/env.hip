// Stand-alone environment kernels for host-driven callers (RandomAgent, user code).
#include "env_device.h"

__global__ __launch_bounds__(256) void env_mask_kernel(vrp_env e, int parity) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= e.B) return;
  const int N = e.N;
  const uint8_t *vis = e.visited + (size_t)b * N;
  int v0 = (lane < N) ? vis[lane] : 1;
  int v1 = (lane + 64 < N) ? vis[lane + 64] : 1;
  double load = (e.kind == VRP_KIND_IRP) ? e.load[b] : 1.0;
  env_fixups_and_mask(e, b, lane, e.cur[b] == e.depot[b], v0, v1, load,
                      e.mask + (size_t)parity * e.B * N);
}

__global__ __launch_bounds__(256) void env_step_kernel(vrp_env e, const int64_t *actions,
                                                       int parity_out, double *reward,
                                                       int32_t *notdone) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= e.B) return;
  const int a = (int)actions[b];
  EnvStepOut o = env_step_wave(e, b, a, lane, e.mask + (size_t)parity_out * e.B * e.N);
  if (lane == 0) {
    reward[b] = -o.dist;  // tsp.py:98
    if (!o.done) *notdone = 1;  // same value from every writer (see flag_notdone, decoder.hip)
  }
}

__global__ __launch_bounds__(256) void env_features_kernel(vrp_env e, float *x, uint8_t *is_depot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = e.B * e.N;
  if (i >= total) return;
  const int b = i / e.N, n = i - b * e.N;
  x[3 * i + 0] = (float)e.pos[2 * (size_t)i];
  x[3 * i + 1] = (float)e.pos[2 * (size_t)i + 1];
  x[3 * i + 2] = (e.kind == VRP_KIND_IRP) ? (float)e.demand[i] : 0.f;
  is_depot[i] = (n == e.depot[b]) ? 1 : 0;
}

// Start-of-episode state in one launch: visited := 0, both mask buffers := 0,
// current_location := depots, load := 1 (tsp.py:150-160,172-174, irp.py:47,184).
__global__ __launch_bounds__(256) void env_reset_kernel(vrp_env e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = e.B * e.N;
  if (i < total) {
    e.visited[i] = 0;
    e.mask[i] = 0;
    e.mask[(size_t)total + i] = 0;
  }
  if (i < e.B) {
    e.cur[i] = e.depot[i];
    if (e.load) e.load[i] = 1.0;
  }
}

static int check_env(const vrp_env *env) {
  VRP_REQUIRE(env != nullptr, "env is NULL");
  VRP_REQUIRE(env->kind >= 0 && env->kind <= 2, "env.kind %d out of range", env->kind);
  VRP_REQUIRE(env->B > 0 && env->N > 0 && env->N <= VRP_MAX_NODES,
              "env shape B=%d N=%d unsupported (N <= %d)", env->B, env->N, VRP_MAX_NODES);
  VRP_REQUIRE(env->pos && env->depot && env->visited && env->mask && env->cur,
              "env has NULL state pointers");
  VRP_REQUIRE(env->kind != VRP_KIND_IRP || (env->demand && env->load),
              "IRP env needs demand and load");
  return 0;
}

extern "C" int vrp_env_mask(const vrp_env *env, int parity, void *stream) {
  if (int r = check_env(env)) return r;
  hipLaunchKernelGGL(env_mask_kernel, dim3((env->B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     *env, parity & 1);
  VRP_CHECK_LAUNCH("env_mask");
  return 0;
}

extern "C" int vrp_env_step(const vrp_env *env, const int64_t *actions, int parity_out,
                            double *reward_f64, int32_t *notdone, void *stream) {
  if (int r = check_env(env)) return r;
  VRP_REQUIRE(actions && reward_f64 && notdone, "vrp_env_step: NULL argument");
  hipLaunchKernelGGL(env_step_kernel, dim3((env->B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     *env, actions, parity_out & 1, reward_f64, notdone);
  VRP_CHECK_LAUNCH("env_step");
  return 0;
}

extern "C" int vrp_env_features(const vrp_env *env, float *x, uint8_t *is_depot, void *stream) {
  if (int r = check_env(env)) return r;
  VRP_REQUIRE(x && is_depot, "vrp_env_features: NULL argument");
  const int total = env->B * env->N;
  hipLaunchKernelGGL(env_features_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, *env, x, is_depot);
  VRP_CHECK_LAUNCH("env_features");
  return 0;
}

extern "C" int vrp_env_reset(const vrp_env *env, void *stream) {
  if (int r = check_env(env)) return r;
  const int total = env->B * env->N;
  hipLaunchKernelGGL(env_reset_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     *env);
  VRP_CHECK_LAUNCH("env_reset");
  return 0;
}
