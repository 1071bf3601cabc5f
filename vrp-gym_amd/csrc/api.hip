// Error reporting and the rollout drivers (R1 of SURVEY.md 8a).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void vrp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *vrp_last_error(void) { return g_err; }
extern "C" int vrp_abi_version(void) { return 2; }

// x (B,N,3) fp32 and is_depot (B,N) u8 live at the tail of the encoder workspace.
static void feature_scratch(void *enc_ws, int B, int N, int hidden, float **x, uint8_t **isd) {
  const size_t R = (size_t)B * N;
  char *p = (char *)enc_ws + vrp_encoder_workspace_bytes(B, N, hidden) - vrp_align_up(R * 12) -
            vrp_align_up(R);
  *x = (float *)p;
  *isd = (uint8_t *)(p + vrp_align_up(R * 12));
}

// Episode accumulators := 0.  A kernel rather than hipMemsetAsync: memset nodes inside a
// captured hipGraph were observed to race with the kernels that follow them (ROCm 7.0
// runtime bundled with torch 2.10), kernels keep stream order.
__global__ void rollout_init_kernel(float *acc_loss, float *acc_logp, int32_t *notdone, int B,
                                    int nflags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { acc_loss[i] = 0.f; acc_logp[i] = 0.f; }
  if (i < nflags) notdone[i] = 0;
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
  for (int t = 0; t < max_steps; ++t)
    if (int r = vrp_decode_step(kind, derived, dw, env, emb, dec_workspace, io, t, max_steps,
                                flags, stream))
      return r;
  return 0;
}

extern "C" int vrp_rollout(int kind, const vrp_encoder_weights *ew, const vrp_decoder_weights *dw,
                           void *derived, const vrp_env *env, int train, int sample, float *emb,
                           void *enc_workspace, void *dec_workspace, const vrp_rollout_io *io,
                           int max_steps, void *stream) {
  VRP_REQUIRE(ew && dw && derived && env && emb && enc_workspace && dec_workspace && io,
              "rollout: NULL argument");
  VRP_REQUIRE(env->kind == kind, "rollout: env.kind=%d but kind=%d", env->kind, kind);
  const int B = env->B, N = env->N;
  const int need = (kind == VRP_KIND_TSP) ? N - 1 : 2 * (N - 1);
  VRP_REQUIRE(max_steps >= need, "rollout: max_steps=%d < %d", max_steps, need);
  // state = env.get_state(): applies the depot fix-ups (tsp.py:106-129)
  if (int r = vrp_env_mask(env, 0, stream)) return r;
  float *x;
  uint8_t *isd;
  feature_scratch(enc_workspace, B, N, ew->hidden, &x, &isd);
  if (int r = vrp_env_features(env, x, isd, stream)) return r;
  // depot_mask: TSP none; VRP = the state's mask column (QUIRK, graph_vrp_agent.py:67);
  // IRP = the is_depot column (graph_irp_agent.py:77-79)
  const uint8_t *dm = (kind == VRP_KIND_TSP) ? nullptr : (kind == VRP_KIND_VRP ? env->mask : isd);
  if (int r = vrp_encoder_forward(ew, train, B, N, x, dm, emb, enc_workspace, stream)) return r;
  if (int r = vrp_decode_prologue(kind, derived, B, N, emb, dec_workspace, stream)) return r;
  // `sample` doubles as the step flags
  return vrp_rollout_steps(kind, derived, dw, env, emb, dec_workspace, io, max_steps,
                           sample & (VRP_STEP_SAMPLE | VRP_STEP_TILE_KERNEL | VRP_STEP_THROUGHPUT_KERNEL),
                           stream);
}
