// Host-side instance sampler, bit-exact with numpy's legacy global stream (E1 of
// SURVEY.md 8a).  The reference draws every instance with three numpy calls per graph:
//   np.random.rand(N,2); np.random.choice(N, size=1, replace=False);
//   np.random.uniform(1, 10, (N,1)) / (0.2449*N + 26.12)
// (gym_vrp/graph/vrp_graph.py:28-43, called B times by vrp_network.py:41-42).  Calling
// them from Python costs ~30 us per graph (0.26 s at B = 8192).  This file replays the
// same MT19937 stream natively: the caller hands over numpy's generator state
// (np.random.get_state()), the stream is advanced exactly as numpy would, and the state
// goes back with np.random.set_state().  Algorithms restated from numpy's legacy
// RandomState: random_sample = genrand_res53, choice(replace=False) = permutation =
// Fisher-Yates from the top with the masked-rejection random_interval (32-bit draws),
// uniform = low + (high-low)*random_sample.
#include <stdint.h>
#include "common.h"

namespace {
struct MT {
  uint32_t *mt;
  int pos;
  inline uint32_t next() {
    constexpr int N = 624, M = 397;
    if (pos >= N) {
      int kk = 0;
      uint32_t y;
      for (; kk < N - M; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      for (; kk < N - 1; ++kk) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
      mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  inline double sample() {  // genrand_res53
    const uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  inline uint32_t interval(uint32_t mx) {  // uniform on [0, mx], masked rejection
    if (mx == 0) return 0;
    uint32_t mask = mx;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (next() & mask)) > mx) {}
    return v;
  }
};
}  // namespace

// key: numpy's 624-word MT19937 state (in/out); *pos: its position (in/out).
// pos_out (B,N,2) f64, depots (B) i64, demands (B,N) f64.  Host pointers.
// Graphs [first, first + count) of the B the stream is advanced over are stored (outputs hold
// `count` graphs); the others are drawn and discarded -- MT19937 has no cheap skip-ahead and the
// permutation's rejection sampling makes the number of draws per graph data-dependent, so a rank
// of a sharded env replays the whole stream natively (~1 us per graph) but keeps only its rows.
extern "C" int vrp_draw_instances_host_range(uint32_t *key_host, int32_t *pos_host, int B, int N,
                                             int first, int count, double *pos_out_host,
                                             int64_t *depots_host, double *demands_host) {
  VRP_REQUIRE(key_host && pos_host && pos_out_host && depots_host && demands_host,
              "draw_instances: NULL argument");
  VRP_REQUIRE(B > 0 && N > 0 && N <= 65536, "draw_instances: bad shape B=%d N=%d", B, N);
  VRP_REQUIRE(first >= 0 && count >= 0 && first + count <= B,
              "draw_instances: bad range first=%d count=%d of %d", first, count, B);
  VRP_REQUIRE(*pos_host >= 0 && *pos_host <= 624, "draw_instances: bad generator position");
  MT g{key_host, *pos_host};
  const double scale = 0.2449 * N + 26.12;  // vrp_graph.py:41
  uint32_t *perm = new uint32_t[N];
  for (int b = 0; b < B; ++b) {
    const bool keep = b >= first && b < first + count;
    double *p = pos_out_host + (size_t)(b - first) * N * 2;
    if (keep) for (int i = 0; i < 2 * N; ++i) p[i] = g.sample();  // rand(N,2)
    else for (int i = 0; i < 4 * N; ++i) (void)g.next();          //   (two words per sample)
    for (int i = 0; i < N; ++i) perm[i] = (uint32_t)i;            // permutation(N)[:1]
    for (int i = N - 1; i > 0; --i) {
      const uint32_t j = g.interval((uint32_t)i);
      const uint32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    if (!keep) {
      for (int i = 0; i < 2 * N; ++i) (void)g.next();             // uniform(1,10,(N,1))
      continue;
    }
    depots_host[b - first] = perm[0];
    double *d = demands_host + (size_t)(b - first) * N;
    for (int i = 0; i < N; ++i) {
      const double u = g.sample();
      const double v = 1.0 + 9.0 * u;                             // uniform(1,10)
      d[i] = v / scale;
    }
    d[perm[0]] = 0.0;                                             // vrp_graph.py:43
  }
  delete[] perm;
  *pos_host = g.pos;
  return 0;
}

extern "C" int vrp_draw_instances_host(uint32_t *key_host, int32_t *pos_host, int B, int N,
                                       double *pos_out_host, int64_t *depots_host,
                                       double *demands_host) {
  return vrp_draw_instances_host_range(key_host, pos_host, B, N, 0, B, pos_out_host, depots_host,
                                       demands_host);
}

// ------------------------------------------------------------------ device-side sampler
// Throughput runs at large batch: same distributions as the
// reference (coordinates U[0,1)^2, one uniformly chosen depot, demand U[1,10)/(0.2449N+26.12)
// with the depot's forced to 0, vrp_graph.py:28-43) but NOT its random stream -- a
// counter-based Philox4x32-10 keyed by the seed, counter = (global graph index, node,
// episode, draw).  No state, no host work, no upload; a sharded run (first_graph = the
// rank's offset) draws exactly the instances of the unsharded one.
namespace {
// 53 random bits -> [0,1), like numpy's random_sample
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
}  // namespace

__global__ __launch_bounds__(256) void draw_instances_kernel(uint64_t seed, uint64_t episode,
                                                             int first_graph, int B, int N,
                                                             double *__restrict__ pos,
                                                             int32_t *__restrict__ depot,
                                                             double *__restrict__ demand) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N, n = i - b * N;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t g = (uint32_t)(first_graph + b), e0 = (uint32_t)episode;
  const uint32_t e1 = (uint32_t)(episode >> 32) << 2;  // low two bits of word 3 = draw id
  uint32_t c[4] = {g, (uint32_t)n, e0, e1 | 0u};
  vrp_philox4x32_10(c, k0, k1);
  reinterpret_cast<double2 *>(pos)[i] = make_double2(u53(c[0], c[1]), u53(c[2], c[3]));
  uint32_t d[4] = {g, (uint32_t)n, e0, e1 | 1u};
  vrp_philox4x32_10(d, k0, k1);
  uint32_t q[4] = {g, 0xFFFFFFFFu, e0, e1 | 2u};
  vrp_philox4x32_10(q, k0, k1);
  int dep = (int)(u53(q[0], q[1]) * (double)N);
  if (dep >= N) dep = N - 1;
  demand[i] = (n == dep) ? 0.0 : (1.0 + 9.0 * u53(d[0], d[1])) / (0.2449 * (double)N + 26.12);
  if (n == 0) depot[b] = dep;
}

extern "C" int vrp_draw_instances_device(uint64_t seed, uint64_t episode, int first_graph, int B,
                                         int N, double *pos, int32_t *depot, double *demand,
                                         void *stream) {
  VRP_REQUIRE(pos && depot && demand, "draw_instances_device: NULL argument");
  VRP_REQUIRE(B > 0 && N >= 1 && N <= VRP_MAX_NODES && first_graph >= 0,
              "draw_instances_device: bad shape B=%d N=%d first=%d", B, N, first_graph);
  const long total = (long)B * N;
  hipLaunchKernelGGL(draw_instances_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, seed, episode, first_graph, B, N, pos, depot, demand);
  VRP_CHECK_LAUNCH("draw_instances");
  return 0;
}

// ------------------------------------------------------------------ device-side random policy
// RandomAgent.forward (agents/random_agent.py:15-41) for throughput runs: per step and graph
// one node drawn uniformly among the unmasked ones, env.step on it, fp32 reward accumulation.
// The draw comes from the Philox stream (counter = graph, step, episode), NOT from the
// reference's global numpy stream, so tours differ from the host RandomAgent's; the
// distribution is the same.  One wave per graph, same env code as every other kernel.
#include "env_device.h"

__global__ __launch_bounds__(256) void random_step_kernel(vrp_env e, uint64_t seed,
                                                          uint64_t episode, int first_graph, int t,
                                                          float *__restrict__ acc_loss,
                                                          int32_t *__restrict__ notdone,
                                                          int64_t *__restrict__ actions) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= e.B) return;
  if (t > 0 && notdone[t - 1] == 0) return;  // batch-wide done (tsp.py:95): exact no-op
  const int N = e.N, par = t & 1;
  const uint8_t *mask_in = e.mask + (size_t)par * e.B * N + (size_t)b * N;
  const bool s0 = lane < N && mask_in[lane] == 0;
  const bool s1 = lane + 64 < N && mask_in[lane + 64] == 0;
  const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1);
  const int c0 = __popcll(m0), cnt = c0 + __popcll(m1);  // >= 1: a feasible action always exists
  uint32_t c[4] = {(uint32_t)(first_graph + b), (uint32_t)t, (uint32_t)episode,
                   ((uint32_t)(episode >> 32) << 2) | 3u};
  vrp_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  int k = (int)(u53(c[0], c[1]) * (double)cnt);
  if (k >= cnt) k = cnt - 1;
  unsigned long long bits = k < c0 ? m0 : m1;
  int kk = k < c0 ? k : k - c0;
  for (int i = 0; i < kk; ++i) bits &= bits - 1;
  const int a = (k < c0 ? 0 : 64) + __ffsll((long long)bits) - 1;
  EnvStepOut o = env_step_wave(e, b, a, lane, e.mask + (size_t)(par ^ 1) * e.B * N);
  if (lane == 0) {
    acc_loss[b] = (t == 0 ? 0.f : acc_loss[b]) + (float)(-o.dist);  // random_agent.py:39
    if (!o.done) notdone[t] = 1;
    if (actions) actions[(size_t)t * e.B + b] = a;
  }
}

__global__ void random_init_kernel(int32_t *notdone, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) notdone[i] = 0;
}

extern "C" int vrp_random_rollout(const vrp_env *env, uint64_t seed, uint64_t episode,
                                  int first_graph, int max_steps, float *acc_loss,
                                  int32_t *notdone, int64_t *actions, void *stream) {
  VRP_REQUIRE(env && acc_loss && notdone, "random_rollout: NULL argument");
  VRP_REQUIRE(env->B > 0 && env->N >= 2 && env->N <= VRP_MAX_NODES && max_steps > 0,
              "random_rollout: bad shape B=%d N=%d steps=%d", env->B, env->N, max_steps);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(random_init_kernel, dim3((max_steps + 256) / 256), dim3(256), 0, st, notdone,
                     max_steps + 1);
  VRP_CHECK_LAUNCH("random_init");
  if (int r = vrp_env_mask(env, 0, stream)) return r;  // state = env.get_state()
  for (int t = 0; t < max_steps; ++t) {
    hipLaunchKernelGGL(random_step_kernel, dim3((env->B + 3) / 4), dim3(256), 0, st, *env, seed,
                       episode, first_graph, t, acc_loss, notdone, actions);
    VRP_CHECK_LAUNCH("random_step");
  }
  return 0;
}

// ---- test hook: the bits -> Exp(1) map of the in-kernel sampler (common.h), so that a test can
// feed it the extreme draws (0, 0xFFFFFFFF) and assert 0 < q < inf
__global__ void exp1_from_bits_kernel(const uint32_t *__restrict__ bits, float *__restrict__ out,
                                      int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vrp_exp1_from_bits(bits[i]);
}
extern "C" int vrp_debug_exp1_from_bits(const uint32_t *bits, float *out, int n, void *stream) {
  VRP_REQUIRE(bits && out && n > 0, "debug_exp1_from_bits: bad argument");
  hipLaunchKernelGGL(exp1_from_bits_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, bits, out, n);
  VRP_CHECK_LAUNCH("exp1_from_bits");
  return 0;
}
