// Error reporting and the rollout drivers (R1 of SURVEY.md 8a).
#include <stdarg.h>
#include "decoder_step.h"

static thread_local char g_err[512] = "";

void vrp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *vrp_last_error(void) { return g_err; }
extern "C" int vrp_abi_version(void) { return VRP_ABI_VERSION; }
// sha256 (first 16 hex digits) over the kernel sources this library was built from, set by the
// Makefile: measurements kept under profiles/ (PMC traffic) carry it, and bench.py reports them
// only for the build they were taken on
#ifndef VRP_SOURCE_SHA
#define VRP_SOURCE_SHA "unknown"
#endif
extern "C" const char *vrp_source_hash(void) { return VRP_SOURCE_SHA; }

// Episode accumulators := 0.  A kernel rather than hipMemsetAsync: memset nodes inside a
// captured hipGraph were observed to race with the kernels that follow them (ROCm 7.0
// runtime bundled with torch 2.10), kernels keep stream order.
__global__ void rollout_init_kernel(float *acc_loss, float *acc_logp, int32_t *notdone, int B,
                                    int nflags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { acc_loss[i] = 0.f; acc_logp[i] = 0.f; }
  if (i < nflags) notdone[i] = 0;
}

// Host-side pacing of the per-step launch loop (vrp_rollout_steps_range): four pinned flag words
// and their events per host thread and device.
#define STEP_CHUNK 8
struct StepThrottle {
  int32_t *flags = nullptr;   // hipHostMalloc: [4]
  hipEvent_t ev[4] = {};
  int dev = -1;
};
static StepThrottle *step_throttle(hipStream_t st) {
  if (getenv("VRP_NO_THROTTLE")) return nullptr;   // read per call: tests and A/B runs flip it
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VRP_MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
  static thread_local StepThrottle per_dev[VRP_MAX_DEVICES];
  StepThrottle &t = per_dev[dev];
  if (t.dev != dev) {
    if (hipHostMalloc((void **)&t.flags, 4 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&t.ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    t.dev = dev;
  }
  return &t;
}

extern "C" int vrp_rollout_steps_range(int kind, const void *derived,
                                       const vrp_decoder_weights *dw, const vrp_env *env,
                                       const float *emb, void *dec_workspace,
                                       const vrp_rollout_io *io, int t_begin, int t_end,
                                       int max_steps, int flags, void *stream) {
  VRP_REQUIRE(0 <= t_begin && t_begin <= t_end && t_end <= max_steps,
              "rollout_steps_range: [%d,%d) outside [0,%d]", t_begin, t_end, max_steps);
  VRP_REQUIRE(derived && env && emb && dec_workspace && io, "rollout_steps_range: NULL argument");
  const int pwaves = (t_end == max_steps && t_end - t_begin >= 2)
                         ? vrp_persistent_width(kind, env->B, env->N, max_steps, flags, io,
                                                (hipStream_t)stream) : 0;
  if (pwaves > 0) {
    // latency-bound regime: step 0 as its own launch, every later step inside ONE persistent
    // launch (decoder_persistent.hip), one, two or four waves per graph.  Two- and four-wave
    // grids of small batches compute the first node's part of the score rows themselves (no
    // first-node GEMM, no score_base launch behind step 0); a four-wave TSP grid also finalizes
    // itself: four launches per rollout with the encoder and the prologue
    bool fold_first = false;
    if (t_begin == 0) {
      fold_first = vrp_persistent_folds_first(kind, env->B, env->N, pwaves, flags);
      if (int r = vrp_decode_step(kind, derived, dw, env, emb, dec_workspace, io, 0, max_steps,
                                  flags | (fold_first ? VRP_STEP_NO_FIRST_ROW : 0), stream)) return r;
      t_begin = 1;
    }
    VRP_REQUIRE(env->kind == kind && io->acc_loss && io->acc_logp && io->notdone,
                "rollout_steps_range: bad env/io");
    VRP_REQUIRE(!(flags & VRP_STEP_SAMPLE) || io->noise || io->noise_seed,
                "rollout_steps_range: sampling needs io.noise or io.noise_seed");
    const StepParams sp = vrp_make_step_params(kind, derived, env, emb, dec_workspace, io, t_begin,
                                                max_steps, flags);
    return vrp_launch_persistent_steps(sp, dec_workspace, (hipStream_t)stream, pwaves,
                                       fold_first ? derived : nullptr);
  }
  // One launch per step.  A VRP / IRP episode is over after anything between N - 1 and 2 (N - 1)
  // steps (tsp.py:95: the batch-wide done flag), and a launch queued behind the end leaves at its
  // first instruction -- but still costs its 3-4 us (VRP-100 x 2048 sampling: 86 of 198 launches,
  // 0.35 ms per rollout).  So from step N - 1 on the host stays at most two chunks of STEP_CHUNK
  // launches ahead of the device: the flag of a chunk's last step comes back through a pinned
  // word, and the loop stops at the first chunk the device reports finished (at most
  // 2 STEP_CHUNK - 1 empty launches).  Off while the stream is being captured into a hipGraph
  // (the fixed-length loop is what a graph needs) and with VRP_NO_THROTTLE set (A/B).
  StepThrottle *th = nullptr;
  const int tmin = env->N - 1;
  if (kind != VRP_KIND_TSP && t_end == max_steps && t_begin <= tmin &&
      max_steps - tmin >= 2 * STEP_CHUNK && io->notdone && !(flags & VRP_STEP_DECODE_ONLY))
    th = step_throttle((hipStream_t)stream);
  int queued = 0;   // checkpoints recorded so far
  for (int t = t_begin; t < t_end; ++t) {
    if (th && t >= tmin + 2 * STEP_CHUNK && (t - tmin) % STEP_CHUNK == 0) {
      // checkpoint i = (t - tmin) / STEP_CHUNK - 2 covers step tmin + STEP_CHUNK (i + 1) - 1
      const int i = (t - tmin) / STEP_CHUNK - 2;
      if (i < queued) {
        if (hipEventSynchronize(th->ev[i % 4]) != hipSuccess) { (void)hipGetLastError(); th = nullptr; }
        else if (th->flags[i % 4] == 0) break;   // that step found every graph finished
      }
    }
    if (int r = vrp_decode_step(kind, derived, dw, env, emb, dec_workspace, io, t, max_steps,
                                flags, stream))
      return r;
    if (th && t >= tmin && (t - tmin + 1) % STEP_CHUNK == 0) {
      const int i = queued % 4;
      if (hipMemcpyAsync(&th->flags[i], io->notdone + t, sizeof(int32_t), hipMemcpyDeviceToHost,
                         (hipStream_t)stream) != hipSuccess ||
          hipEventRecord(th->ev[i], (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        th = nullptr;
      } else {
        ++queued;
      }
    }
  }
  return 0;
}

static int rollout_step_loop(int kind, const void *derived, const vrp_decoder_weights *dw,
                             const vrp_env *env, const float *emb, void *dec_workspace,
                             const vrp_rollout_io *io, int max_steps, int flags, void *stream) {
  return vrp_rollout_steps_range(kind, derived, dw, env, emb, dec_workspace, io, 0, max_steps,
                                 max_steps, flags, stream);
}

extern "C" int vrp_rollout_steps(int kind, const void *derived, const vrp_decoder_weights *dw,
                                 const vrp_env *env, const float *emb, void *dec_workspace,
                                 const vrp_rollout_io *io, int max_steps, int flags,
                                 void *stream) {
  VRP_REQUIRE(io && io->acc_loss && io->acc_logp && io->notdone, "rollout: io NULL");
  hipStream_t st = (hipStream_t)stream;
  const int B = env->B;
  const int n = (B > max_steps + 1) ? B : max_steps + 1;
  hipLaunchKernelGGL(rollout_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, io->acc_loss,
                     io->acc_logp, io->notdone, B, max_steps + 1);
  VRP_CHECK_LAUNCH("rollout_init");
  return rollout_step_loop(kind, derived, dw, env, emb, dec_workspace, io, max_steps, flags, stream);
}

int vrp_encoder_forward_from_env(const vrp_encoder_weights *w, int train, const vrp_env *env,
                                 float *emb, void *workspace, float *acc_loss, float *acc_logp,
                                 int32_t *notdone, int nflags, const float *dec_mb, float *dec_g,
                                 float *dec_cvec, unsigned long long *dec_hist, int32_t *dec_err,
                                 const float *dec_warm, int dec_warm_floats,
                                 const float *dec_wqgT, const float *dec_bq, float *dec_QG,
                                 int *decoder_constants_done, hipStream_t st);
int vrp_decode_prologue_ex(int kind, const void *derived, int B, int N, const float *emb,
                           void *workspace, int constants_done, void *stream);

// The encoder phase of vrp_rollout: state = env.get_state() (mask with the depot fix-ups,
// tsp.py:106-129), the network inputs, the embedding and the zeroed accumulators; depot_mask: TSP
// none, VRP = the state's mask column (QUIRK graph_vrp_agent.py:67), IRP = is_depot
// (graph_irp_agent.py:77-79).  Small eval-mode batches: ONE launch (encoder_stack_kernel) that
// also leaves the decoder's per-graph constants (*constants_done = 1).
static int rollout_encode(int kind, const vrp_encoder_weights *ew, void *derived,
                          const vrp_env *env, int train, float *emb, void *enc_workspace,
                          void *dec_workspace, const vrp_rollout_io *io, int max_steps,
                          int *constants_done, void *stream) {
  VRP_REQUIRE(ew && derived && env && emb && enc_workspace && dec_workspace && io,
              "rollout: NULL argument");
  VRP_REQUIRE(io->acc_loss && io->acc_logp && io->notdone, "rollout: io accumulators NULL");
  VRP_REQUIRE(env->kind == kind, "rollout: env.kind=%d but kind=%d", env->kind, kind);
  VRP_REQUIRE(env->pos && env->depot && env->visited && env->mask && env->cur,
              "rollout: env has NULL state pointers");
  VRP_REQUIRE(kind != VRP_KIND_IRP || (env->demand && env->load), "rollout: IRP needs demand/load");
  const int B = env->B, N = env->N;
  const int need = (kind == VRP_KIND_TSP) ? N - 1 : 2 * (N - 1);
  VRP_REQUIRE(max_steps >= need, "rollout: max_steps=%d < %d", max_steps, need);
  Derived d = carve_derived(derived);
  DecWs w = carve_decws(dec_workspace, B, N);
  return vrp_encoder_forward_from_env(ew, train, env, emb, enc_workspace, io->acc_loss,
                                      io->acc_logp, io->notdone, max_steps + 1, d.mb, w.g, w.cvec,
                                      w.hist, w.err, use_fused_prologue(N) ? d.Wproj : nullptr,
                                      1536 * 128, d.WqgT, d.bq, w.QG, constants_done,
                                      (hipStream_t)stream);
}

extern "C" int vrp_rollout_encode(int kind, const vrp_encoder_weights *ew, void *derived,
                                  const vrp_env *env, int train, float *emb, void *enc_workspace,
                                  void *dec_workspace, const vrp_rollout_io *io, int max_steps,
                                  void *stream) {
  int constants_done = 0;
  return rollout_encode(kind, ew, derived, env, train, emb, enc_workspace, dec_workspace, io,
                        max_steps, &constants_done, stream);
}

extern "C" int vrp_rollout(int kind, const vrp_encoder_weights *ew, const vrp_decoder_weights *dw,
                           void *derived, const vrp_env *env, int train, int sample, float *emb,
                           void *enc_workspace, void *dec_workspace, const vrp_rollout_io *io,
                           int max_steps, void *stream) {
  VRP_REQUIRE(dw, "rollout: NULL argument");
  int constants_done = 0;
  if (int r = rollout_encode(kind, ew, derived, env, train, emb, enc_workspace, dec_workspace, io,
                             max_steps, &constants_done, stream)) return r;
  const int B = env->B, N = env->N;
  if (int r = vrp_decode_prologue_ex(kind, derived, B, N, emb, dec_workspace, constants_done,
                                     stream)) return r;
  // `sample` doubles as the step flags
  return rollout_step_loop(kind, derived, dw, env, emb, dec_workspace, io, max_steps,
                           sample & (VRP_STEP_SAMPLE | VRP_STEP_TILE_KERNEL | VRP_STEP_TABLE_KERNEL |
                                     VRP_STEP_THROUGHPUT_KERNEL | VRP_STEP_NO_PERSISTENT), stream);
}
