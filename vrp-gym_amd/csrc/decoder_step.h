// Shared by the per-step decode kernels (decoder.hip) and the persistent multi-step one
// (decoder_persistent.hip): launch parameters and the row-pipeline helpers.
#pragma once
#include "env_device.h"
#include "decoder_ws.h"

#define VRP_MAX_DEVICES 64   // per-device host-side state (residency census, launch pacing)
struct StepParams {
  int kind, B, N, t, max_steps, sample, decode_only;
  const float *emb;
  const float *embP;            // row-paired copy of emb (DecWs::embP) or NULL
  const float *row0, *SLD, *SL, *base;
  float *curs;
  int32_t *last, *first;
  const float *WvT, *bv, *MT, *mb;
  const float *Wv, *M;          // (384,128) v_proj rows; (128,384) M = Wkp^T Watt Wo / sqrt(128)
  const float *WvP, *MP;        // the same two in MFMA fragment order (decoder_ws.h)
  const float *RT, *cvec;
  float clip;                   // C of graph_decoder.py:56,97 (10 unless io.logit_clip says otherwise)
  int dbg;                      // tuning aid (VRP_TILE_DBG): stop the tile kernel after phase dbg
  int stagger;                  // raw-tile kernel: odd workgroups start this many x 64 cycles late
  int skip_curs;                // raw-tile kernel: the next step is the raw-tile kernel's too (the
                                // host's per-step schedule), `curs` need not be kept current
  vrp_env env;
  vrp_rollout_io io;
};

#ifndef RT_U
#define RT_U 5  // float4 loads per lane per work item (N = 40: 10 per row share -> 2 items)
#endif
#ifndef RT_NB
#define RT_NB 3  // work items in flight per wave, large-batch mode (114 VGPRs: four waves per SIMD)
#endif
#ifndef RT_MINW
#define RT_MINW 3
#endif

// lane owns float4 indices part + 8*i (i < cnt) of its row; one work item = RT_U of them
__device__ __forceinline__ void rt_load(float4 (&r)[RT_U], const float4 *rt, int i0, int cnt,
                                        bool on) {
#pragma unroll
  for (int i = 0; i < RT_U; ++i)
    r[i] = (on && i0 + i < cnt) ? rt[8 * (i0 + i)] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float rt_dot(float acc, const float4 (&r)[RT_U], const float4 *a,
                                        int i0, int cnt) {
#pragma unroll
  for (int i = 0; i < RT_U; ++i) {
    if (i0 + i < cnt) {
      const float4 w = a[8 * (i0 + i)];
      acc = fmaf(w.x, r[i].x, acc);
      acc = fmaf(w.y, r[i].y, acc);
      acc = fmaf(w.z, r[i].z, acc);
      acc = fmaf(w.w, r[i].w, acc);
    }
  }
  return acc;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const long long x = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)x, l);
  const int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// index of the k-th set bit of `bits` (k < popcount), wave-uniform inputs per lane group
__device__ __forceinline__ int kth_set_bit(unsigned long long bits, int k) {
  for (int i = 0; i < k; ++i) bits &= bits - 1;
  return __ffsll((long long)bits) - 1;
}


StepParams vrp_make_step_params(int kind, const void *derived, const vrp_env *env, const float *emb,
                                void *workspace, const vrp_rollout_io *io, int t, int max_steps,
                                int flags);
bool vrp_persistent_eligible(int kind, int B, int N, int max_steps, int flags,
                             const vrp_rollout_io *io, hipStream_t st);
int vrp_launch_persistent_steps(const StepParams &sp, void *workspace, hipStream_t st,
                                int waves = 1, const void *derived_for_first = nullptr);
bool vrp_persistent_folds_first(int kind, int B, int N, int waves, int flags);

int vrp_persistent_width(int kind, int B, int N, int max_steps, int flags, const vrp_rollout_io *io,
                         hipStream_t st);
void vrp_persistent_serialize_begin(hipStream_t st, void **token);
void vrp_persistent_serialize_end(hipStream_t st, void *token);
bool vrp_tile_mfma_supported(int N);
int vrp_launch_tile_mfma_step(const StepParams &p, hipStream_t st);
bool vrp_tile2_supported(int N);
int vrp_usable_cus(hipStream_t st);   // compute units the residency census found usable (decoder_persistent.hip)
int vrp_launch_tile2_step(const StepParams &p, hipStream_t st);
