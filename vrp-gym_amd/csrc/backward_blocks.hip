// Building blocks of the hand-written backward pass (K4 of SURVEY.md 7.1): the gradient of
// sum_t log p(a_t) w.r.t. every parameter (REINFORCE, agents/graph_tsp_agent.py:178-186).
//   gemm_tn     dW = dY^T X        (weight gradients; fp32 MFMA, deterministic split-K)
//   colsum      db = sum_r dY[r]   (bias gradients)
//   transpose   W^T copies so that dX = dY W runs on the forward GEMM kernel
//   bn_bwd      BatchNorm1d backward with batch statistics (train mode)
//   attn_bwd    per-(graph, head) softmax-attention backward of the encoder
//   embed_bwd   node/depot embedding weight gradients
// Everything is fp32; reductions over rows are done in a fixed order (no float atomics), so
// gradients are bitwise reproducible.
#include "common.h"
#include "x3_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ C = X^T Y (split over rows)
// X (R,N1), Y (R,N2) row-major.  Workgroup (tile i, tile j, split s) computes the 128x128
// tile of X[r0:r1]^T Y[r0:r1] with v_mfma_f32_32x32x2_f32 (A[i][k=r] = X[r][i]: lanes run
// along i, so the LDS reads of a row-major (r, col) tile are conflict-free) and writes it to
// slab s; a second kernel sums the slabs in order.
#ifndef TN_BR
#define TN_BR 32
#endif
// measured (tools/train_probe.py, VRP-40 x 2048 epoch, 18 products of 81920 rows): two workgroups
// per CU and 512 of them 124 us per product, three and 768: 117, four and 1024: 116; 64-row slabs
// (half the barriers) 127
#ifndef TN_MINWG
#define TN_MINWG 4
#endif
#ifndef TN_TARGET
#define TN_TARGET 512    // workgroups per launch (tiles x row splits): two per CU (round 4: 1024 wrote twice the slabs)
#endif
__global__ __launch_bounds__(256, TN_MINWG) void gemm_tn_kernel(const float *__restrict__ X, int ldx,
                                                         const float *__restrict__ Y, int ldy,
                                                         float *__restrict__ slabs, int R, int N1,
                                                         int N2, int rows_per_split) {
  __shared__ float4 Xs[TN_BR * 32];  // [r][128 cols] as float4
  __shared__ float4 Ys[TN_BR * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.x * 128, j0 = blockIdx.y * 128, sp = blockIdx.z;
  const int r_begin = sp * rows_per_split;
  const int r_end = min(R, r_begin + rows_per_split);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int lr = tid >> 5, lc = tid & 31;  // 8 rows x 32 float4 columns per pass
  const int fi = lane & 31, fk = lane >> 5;
  // the next 32-row slab travels in registers while the current one feeds the MFMAs
  float4 xv[TN_BR / 8], yv[TN_BR / 8];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int s = 0; s < TN_BR / 8; ++s) {
      const int r = r0 + lr + 8 * s;
      xv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      yv[s] = xv[s];
      if (r < r_end) {
        xv[s] = *reinterpret_cast<const float4 *>(X + (size_t)r * ldx + i0 + 4 * lc);
        yv[s] = *reinterpret_cast<const float4 *>(Y + (size_t)r * ldy + j0 + 4 * lc);
      }
    }
  };
  fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += TN_BR) {
#pragma unroll
    for (int s = 0; s < TN_BR / 8; ++s) {
      Xs[(lr + 8 * s) * 32 + lc] = xv[s];
      Ys[(lr + 8 * s) * 32 + lc] = yv[s];
    }
    __syncthreads();
    if (r0 + TN_BR < r_end) fetch(r0 + TN_BR);
    const float *xs = reinterpret_cast<const float *>(Xs);
    const float *ys = reinterpret_cast<const float *>(Ys);
#pragma unroll
    for (int rr = 0; rr < TN_BR; rr += 2) {
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = xs[(rr + fk) * 128 + wm * 64 + t * 32 + fi];
        b[t] = ys[(rr + fk) * 128 + wn * 64 + t * 32 + fi];
      }
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }
  float *out = slabs + (size_t)sp * N1 * N2;
  const int col = lane & 31, rq = lane >> 5;
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wm * 64 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
        const int j = j0 + wn * 64 + tb * 32 + col;
        out[(size_t)i * N2 + j] = acc[ta][tb][r];
      }
}

// ---- the same product on the bf16 matrix cores (round 5; x3_common.h): every fp32 operand as three
// bf16 planes, six v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation -- the fp32 MFMA's
// accuracy (tools/gemm_tn_probe.py: relative error 5e-7 either way).  The inner dimension of this
// product is the ROW index, so an MFMA operand is eight consecutive rows of one column: a thread
// fetches rows r .. r + 7 of its four columns (eight 16-byte loads; a wave's load instruction covers
// two whole 512-byte rows), which IS the 8-value chunk of each of those columns -- the
// transposition happens in registers, for free.  It splits the four chunks and stores the planes as
// 16-byte pieces: LDS image [plane][chunk of 8 rows][slot], slot(col) = (col & 3) * 32 +
// ((col >> 2) + 4 (col & 3)) % 32 -- the store instruction of component c writes 32 consecutive
// slots, and the 16 columns of an operand read fall on 16 different 16-byte bank groups
// (SQ_LDS_BANK_CONFLICT = 0).  Waves 2 x 2, a wave owns 4 x 4 tiles of 16 x 16 (64 accumulator
// registers); a 32-row slab = one MFMA k-step: 24 operand reads, 96 MFMAs per wave; two workgroups
// per CU.  Measured on one box (81920 rows, product + slab sum): 128 x 512 116 -> 82 us, 384 x 128
// 96 -> 70 us, 384 x 384 (102400 rows) 328 -> 211 us: 130-143 fp32-equivalent TFLOP/s.  Ablation at
// 128 x 512 (82 us with the 9 us slab sum): without the splits 70, without the loads 79, without
// both 67, without the MFMAs 23 -- the MFMA phase of a slab (24 operand reads behind a barrier, 96
// MFMAs, a barrier) runs at under half the pipe's rate (26 us of MFMA issue at 16 cycles each,
// tools/micro/mfma_rate): per slab pair a CU's LDS is busy 2.3 k cycles next to the matrix pipe's
// 3.1 k, and the two resident workgroups overlap the two only partly.  Built and measured, not
// kept: two LDS buffers with the next slab's split between the MFMAs (one workgroup per CU, one
// barrier per slab: 93 us), the tiles of a row split on one XCD (no change: the repeated reads
// already hit the last-level cache), three workgroups per CU (spills: 143 us).
#define TNX_PLANE (4 * 128 * 8)   // bf16 per plane of a 32-row x 128-column block
__device__ __forceinline__ int tnx_slot(int col) { return (col & 3) * 32 + (((col >> 2) + 4 * (col & 3)) & 31); }
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(const float *__restrict__ X, int ldx,
                                                            const float *__restrict__ Y, int ldy,
                                                            float *__restrict__ slabs, int R, int N1,
                                                            int N2, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) __bf16 Ps[2 * 3 * TNX_PLANE];   // X planes, Y planes: 48 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i16 = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * 128, j0 = blockIdx.y * 128, sp = blockIdx.z;
  const int r_begin = sp * rows_per_split;
  const int r_end = min(R, r_begin + rows_per_split);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader role: matrix (X / Y), chunk of 8 rows, group of 4 columns
  const int which = tid >> 7, chunk = (tid >> 5) & 3, cg = tid & 31;
  const float *src = which ? Y + j0 + 4 * cg : X + i0 + 4 * cg;
  const int lds = which ? ldy : ldx;
  // Two slabs ahead: a slab's rows are requested two stages before the stage that splits them.
  // The loads are UNCONDITIONAL (row index clamped, rows outside the split zeroed on arrival):
  // behind a branch the compiler cannot count what is in flight and waits for everything, i.e. for
  // the slab requested a moment ago (92 us instead of 82 at 128 x 512).
  float4 v0[8], v1[8];
  auto fetch = [&](float4 (&v)[8], int r0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + 8 * chunk + e;
      v[e] = *reinterpret_cast<const float4 *>(src + (size_t)min(r, R - 1) * lds);
    }
  };
  auto mask = [&](float4 (&v)[8], int r0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (r0 + 8 * chunk + e >= r_end) v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  bf16x8 *const mine = reinterpret_cast<bf16x8 *>(Ps + which * 3 * TNX_PLANE) + chunk * 128;
  auto store = [&](const float4 (&v)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x8[e] = c == 0 ? v[e].x : c == 1 ? v[e].y : c == 2 ? v[e].z : v[e].w;
      bf16x8 h, m, l;
      x3_split8(x8, h, m, l);
      const int slot = c * 32 + ((cg + 4 * c) & 31);
      mine[slot] = h;
      mine[slot + TNX_PLANE / 8] = m;
      mine[slot + 2 * (TNX_PLANE / 8)] = l;
    }
  };
  // operand addresses: plane 0 of chunk q, this lane's column of tile t
  const bf16x8 *xa[4], *yb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xa[t] = reinterpret_cast<const bf16x8 *>(Ps) + q * 128 + tnx_slot(wm * 64 + 16 * t + i16);
    yb[t] = reinterpret_cast<const bf16x8 *>(Ps + 3 * TNX_PLANE) + q * 128 + tnx_slot(wn * 64 + 16 * t + i16);
  }
  auto mma = [&]() {
    bf16x8 a[4][3], b[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[t][p] = xa[t][p * (TNX_PLANE / 8)];
        b[t][p] = yb[t][p * (TNX_PLANE / 8)];
      }
    // small terms first: m m', h l', l h', h m', m h', h h' (16 independent accumulators per term)
#define TNX_TERM(PA, PB)                                                       \
  _Pragma("unroll") for (int ta = 0; ta < 4; ++ta)                             \
  _Pragma("unroll") for (int tb = 0; tb < 4; ++tb)                             \
      acc[ta][tb] = X3_MFMA(a[ta][PA], b[tb][PB], acc[ta][tb]);
    TNX_TERM(1, 1) TNX_TERM(0, 2) TNX_TERM(2, 0) TNX_TERM(0, 1) TNX_TERM(1, 0) TNX_TERM(0, 0)
#undef TNX_TERM
  };
  fetch(v0, r_begin);
  fetch(v1, r_begin + 32);
  for (int r0 = r_begin; r0 < r_end; r0 += 64) {
    mask(v0, r0);
    store(v0);
    __syncthreads();
    fetch(v0, r0 + 64);
    mma();
    __syncthreads();
    if (r0 + 32 < r_end) {   // (uniform)
      mask(v1, r0 + 32);
      store(v1);
      __syncthreads();
      fetch(v1, r0 + 96);
      mma();
      __syncthreads();
    }
  }
  // D[i = 4 q + e][j = i16] of tile (ta, tb)
  float *out = slabs + (size_t)sp * N1 * N2;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        out[(size_t)(i0 + wm * 64 + 16 * ta + 4 * q + e) * N2 + j0 + wn * 64 + 16 * tb + i16] = acc[ta][tb][e];
}

// C[i] (+)= sum_s slabs[s][i]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slabs,
                                                          float *__restrict__ C, size_t n,
                                                          int nsplit, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? C[i] : 0.f;
#pragma unroll 8
  for (int k = 0; k < nsplit; ++k) s += slabs[(size_t)k * n + i];
  C[i] = s;
}

// Splits over rows: about TN_TARGET workgroups per launch (tiles x splits: two per CU; a one-tile
// product one per CU) while a split keeps at least 64 rows; the partial tiles (slabs) are summed in
// split order.  Measured (tools/gemm_tn_probe.py, 81920 rows; until round 4: 1024 workgroups, at
// most 128 splits -- a 128 x 128 product on half the CUs, a 384 x 128 one with two workgroups on
// half of them and one on the others):
//   384 x 128   120.7 us -> 97.7      128 x 128   67.6 -> 46.7      128 x 384 (102400 rows)  135.0 -> 114.2
//   512 x 128 / 128 x 512  116.6 -> 117.9 (unchanged: 512 workgroups before and after)
// 256 / 384 / 1024 workgroups for the multi-tile shapes: 103 / 112 / 108 us at 384 x 128 (fewer leave
// latency exposed, more write more slabs: 67 MB at 1024).
#ifndef TN_MAXSPLIT
#define TN_MAXSPLIT 256
#endif
static int tn_splits(int R, int N1, int N2) {
  const int tiles = (N1 / 128) * (N2 / 128);
  int nsplit = TN_TARGET / tiles;   // (floor: tiles x splits never exceeds TN_TARGET slabs)
  const int max_by_rows = (R + 63) / 64;
  if (nsplit > max_by_rows) nsplit = max_by_rows;
  if (nsplit > TN_MAXSPLIT) nsplit = TN_MAXSPLIT;
  if (nsplit < 1) nsplit = 1;
  return nsplit;
}

// An upper bound for EVERY product with at most these many tiles: callers size one workspace for
// their largest shape and run smaller ones in it (tiles x splits <= max(TN_TARGET, tiles) slabs of
// 128 x 128 floats).
extern "C" int64_t vrp_gemm_tn_workspace_bytes(int R, int N1, int N2) {
  (void)R;
  const int tiles = (N1 / 128) * (N2 / 128);
  return (int64_t)(tiles > TN_TARGET ? tiles : TN_TARGET) * 128 * 128 * sizeof(float);
}

int vrp_launch_gemm_tn(const float *X, int ldx, const float *Y, int ldy, float *C, int R, int N1,
                       int N2, int accumulate, void *slab_ws, hipStream_t st) {
  VRP_REQUIRE(R > 0 && N1 % 128 == 0 && N2 % 128 == 0, "gemm_tn: bad shape R=%d N1=%d N2=%d", R,
              N1, N2);
  int nsplit = tn_splits(R, N1, N2);
  int rps = (R + nsplit - 1) / nsplit;
  rps = (rps + TN_BR - 1) / TN_BR * TN_BR;
  nsplit = (R + rps - 1) / rps;
  static const bool fp32 = getenv("VRP_GEMM_FP32") != nullptr;   // A/B aid: the fp32-MFMA kernel
  if (fp32)
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(N1 / 128, N2 / 128, nsplit), dim3(256), 0, st, X, ldx, Y,
                       ldy, (float *)slab_ws, R, N1, N2, rps);
  else
    hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3(N1 / 128, N2 / 128, nsplit), dim3(256), 0, st, X, ldx,
                       Y, ldy, (float *)slab_ws, R, N1, N2, rps);
  VRP_CHECK_LAUNCH("gemm_tn");
  const size_t n = (size_t)N1 * N2;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     (const float *)slab_ws, C, n, nsplit, accumulate);
  VRP_CHECK_LAUNCH("slab_reduce");
  return 0;
}

extern "C" int vrp_gemm_tn(const float *X, int ldx, const float *Y, int ldy, float *C, int R,
                           int N1, int N2, int accumulate, void *slab_ws, void *stream) {
  return vrp_launch_gemm_tn(X, ldx, Y, ldy, C, R, N1, N2, accumulate, slab_ws, (hipStream_t)stream);
}

// ------------------------------------------------------------------ column sums (bias gradients)
// out[c] (+)= sum_r Y[r][c].  Stage 1: workgroup (64 columns, row chunk) -> fp64 partial;
// stage 2: one thread per column sums the chunk partials in order (deterministic).
#define CS_MAX_CHUNKS 512
static int cs_chunks(int R) {
  int s = (R + 63) / 64;
  return s > CS_MAX_CHUNKS ? CS_MAX_CHUNKS : (s < 1 ? 1 : s);
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ Y, int ldy,
                                                             int R, int N, int rows_per_chunk,
                                                             double *__restrict__ partial) {
  __shared__ double sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  double s = 0.0;
  if (c < N) {
#pragma unroll 4
    for (int r = r0 + part; r < r1; r += 4) s += (double)Y[(size_t)r * ldy + c];
  }
  sh[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && c < N)
    partial[(size_t)blockIdx.y * N + c] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// 16 threads per column (a chain of `chunks` dependent fp64 adds per thread took 21 us)
__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ partial,
                                                           int chunks, int N,
                                                           float *__restrict__ out, int accumulate) {
  __shared__ double sh[16][16];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c < N) {
#pragma unroll 4
    for (int k = part; k < chunks; k += 16) s += partial[(size_t)k * N + c];
  }
  sh[part][cl] = s;
  __syncthreads();
  if (part == 0 && c < N) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sh[j][cl];
    out[c] = (accumulate ? out[c] : 0.f) + (float)t;
  }
}

extern "C" int64_t vrp_colsum_workspace_bytes(int R, int N) {
  return (int64_t)cs_chunks(R) * N * sizeof(double);
}

int vrp_launch_colsum(const float *Y, int ldy, int R, int N, float *out, int accumulate, void *ws,
                      hipStream_t st) {
  const int chunks = cs_chunks(R);
  const int rpc = (R + chunks - 1) / chunks;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, st, Y, ldy, R,
                     N, rpc, (double *)ws);
  VRP_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 15) / 16), dim3(256), 0, st,
                     (const double *)ws, chunks, N, out, accumulate);
  VRP_CHECK_LAUNCH("colsum_final");
  return 0;
}

extern "C" int vrp_colsum(const float *Y, int ldy, int R, int N, float *out, int accumulate,
                          void *ws, void *stream) {
  VRP_REQUIRE(Y && out && ws && R > 0 && N > 0, "colsum: bad argument");
  return vrp_launch_colsum(Y, ldy, R, N, out, accumulate, ws, (hipStream_t)stream);
}

// ------------------------------------------------------------------ transpose (small weights)
__global__ void transpose_kernel(const float *__restrict__ src, int rows, int cols, int lds,
                                 float *__restrict__ dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int r = idx / cols, c = idx - r * cols;
  dst[(size_t)c * rows + r] = src[(size_t)r * lds + c];
}

int vrp_launch_transpose(const float *src, int rows, int cols, int lds, float *dst, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, st, src, rows,
                     cols, lds, dst);
  VRP_CHECK_LAUNCH("transpose");
  return 0;
}

// ------------------------------------------------------------------ BatchNorm1d backward (train)
// y = (z - mean) * invstd * gamma + beta over R rows.  Given dy:
//   dbeta = sum dy,  dgamma = sum dy * xhat,
//   dz = gamma * invstd / R * (R * dy - dbeta - xhat * dgamma)
// sums[0:128] = sum dy, sums[128:256] = sum dy*xhat  (fp64, zeroed by the caller)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dy,
                                                            const float *__restrict__ z,
                                                            const float *__restrict__ stats, int R,
                                                            double *__restrict__ sums) {
  __shared__ double sh[2][256];
  const int c = threadIdx.x & 127, par = threadIdx.x >> 7;
  const float mean = stats[c], invstd = stats[128 + c];
  double s0 = 0.0, s1 = 0.0;
  for (int r = blockIdx.x * 2 + par; r < R; r += gridDim.x * 2) {
    const float d = dy[(size_t)r * 128 + c];
    const float xh = (z[(size_t)r * 128 + c] - mean) * invstd;
    s0 += (double)d;
    s1 += (double)d * (double)xh;
  }
  sh[0][threadIdx.x] = s0;
  sh[1][threadIdx.x] = s1;
  __syncthreads();
  if (par == 0) {  // per-block partials go to a slab; summed in order by the apply kernel
    sums[(size_t)blockIdx.x * 256 + c] = sh[0][c] + sh[0][c + 128];
    sums[(size_t)blockIdx.x * 256 + 128 + c] = sh[1][c] + sh[1][c + 128];
  }
}

// 1024 threads: column c = tid & 255, quarter q = tid >> 8 of the block partials; quarters
// are combined in order, so the result does not depend on scheduling
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const double *__restrict__ partial,
                                                               int nblocks,
                                                               float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta,
                                                               double *__restrict__ totals,
                                                               int accumulate) {
  __shared__ double sh[4][256];
  const int c = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int per = (nblocks + 3) / 4;
  const int b0 = q * per, b1 = min(nblocks, b0 + per);
  double s = 0.0;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += partial[(size_t)b * 256 + c];
  sh[q][c] = s;
  __syncthreads();
  if (q != 0) return;
  s = ((sh[0][c] + sh[1][c]) + sh[2][c]) + sh[3][c];
  totals[c] = s;
  if (c < 128) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  else dgamma[c - 128] = (accumulate ? dgamma[c - 128] : 0.f) + (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dy,
                                                           const float *__restrict__ z,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ gamma,
                                                           const double *__restrict__ totals, int R,
                                                           float *__restrict__ dz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * 128) return;
  const int c = (int)(i & 127);
  const float mean = stats[c], invstd = stats[128 + c];
  const float xh = (z[i] - mean) * invstd;
  const float db = (float)totals[c], dg = (float)totals[128 + c];
  dz[i] = gamma[c] * invstd * (dy[i] - (db + xh * dg) / (float)R);
}

extern "C" int64_t vrp_bn_bwd_workspace_bytes(void) { return (int64_t)(1024 + 1) * 256 * 8; }

int vrp_launch_bn_bwd(const float *dy, const float *z, const float *stats, const float *gamma,
                      int R, float *dz, float *dgamma, float *dbeta, int accumulate, void *ws,
                      hipStream_t st) {
  int blocks = (R + 31) / 32;
  if (blocks > 512) blocks = 512;
  double *partial = (double *)ws;
  double *totals = partial + (size_t)1024 * 256;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, st, dy, z, stats, R, partial);
  VRP_CHECK_LAUNCH("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, blocks, dgamma,
                     dbeta, totals, accumulate);
  VRP_CHECK_LAUNCH("bn_bwd_finalize");
  const size_t n = (size_t)R * 128;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, z,
                     stats, gamma, totals, R, dz);
  VRP_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}

extern "C" int vrp_bn_bwd(const float *dy, const float *z, const float *stats, const float *gamma,
                          int R, float *dz, float *dgamma, float *dbeta, int accumulate, void *ws,
                          void *stream) {
  return vrp_launch_bn_bwd(dy, z, stats, gamma, R, dz, dgamma, dbeta, accumulate, ws,
                           (hipStream_t)stream);
}

// ------------------------------------------------------------------ encoder attention backward
// Per (graph, head): P = softmax(Q K^T / 4), O = P V.  Given dO:
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - rowsum(dP o P)),  dQ = dS K / 4,  dK = dS^T Q / 4
// One wave per (graph, head).  Pass A (lane = query row i): row max, row sum, D_i and dQ_i.
// Pass B (lane = key row j): dK_j, dV_j with P recomputed from the row statistics in LDS.
// 16 consecutive floats from a 16-byte aligned LDS address as four 128-bit reads
__device__ __forceinline__ void ld16(float (&v)[16], const float *p) {
#pragma unroll
  for (int d = 0; d < 16; d += 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p + d);
    v[d] = t.x; v[d + 1] = t.y; v[d + 2] = t.z; v[d + 3] = t.w;
  }
}
template <int D>
__device__ __forceinline__ void ldD(float (&v)[D], const float *p) {
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p + d);
    v[d] = t.x; v[d + 1] = t.y; v[d + 2] = t.z; v[d + 3] = t.w;
  }
}

template <int D, int W>
__global__ __launch_bounds__(64 * W) void encoder_attention_bwd_kernel(const float *__restrict__ qkv,
                                                                    const float *__restrict__ dO,
                                                                    float *__restrict__ dqkv,
                                                                    int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, h = blockIdx.y * W + wave;
  const float scale = D == 16 ? 0.25f : (D == 32 ? 0.17677669529663687f : 0.35355339059327373f);
  // per wave: Q,K,V,dO (N x 16 each), row max / inverse sum / D (N each)
  float *Qs = smem + (size_t)wave * ((N * (4 * D + 3) + 3) & ~3);  // 16-byte aligned rows of D floats
  float *Ks = Qs + N * D, *Vs = Ks + N * D, *Gs = Vs + N * D;
  float *mx = Gs + N * D, *isum = mx + N, *Dv = isum + N;
  const float *base = qkv + (size_t)b * N * 384;
  const float *gbase = dO + (size_t)b * N * 128;
  constexpr int D4 = D / 4;
  for (int idx = lane; idx < N * D4; idx += 64) {
    const int j = idx / D4, q4 = (idx % D4) * 4;
    *reinterpret_cast<float4 *>(Qs + j * D + q4) =
        *reinterpret_cast<const float4 *>(base + (size_t)j * 384 + h * D + q4);
    *reinterpret_cast<float4 *>(Ks + j * D + q4) =
        *reinterpret_cast<const float4 *>(base + (size_t)j * 384 + 128 + h * D + q4);
    *reinterpret_cast<float4 *>(Vs + j * D + q4) =
        *reinterpret_cast<const float4 *>(base + (size_t)j * 384 + 256 + h * D + q4);
    *reinterpret_cast<float4 *>(Gs + j * D + q4) =
        *reinterpret_cast<const float4 *>(gbase + (size_t)j * 128 + h * D + q4);
  }
  __syncthreads();
  // ---- pass A: lane = i
  for (int i = lane; i < N; i += 64) {
    float q[D], g[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { q[d] = Qs[i * D + d] * scale; g[d] = Gs[i * D + d]; }  // own row
    float m = -INFINITY;
    for (int j = 0; j < N; ++j) {
      float kk[D];
      ldD<D>(kk, Ks + j * D);
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s = fmaf(q[d], kk[d], s);
      m = fmaxf(m, s);
    }
    float l = 0.f, Dacc = 0.f;
    for (int j = 0; j < N; ++j) {
      float kk[D], vv[D];
      ldD<D>(kk, Ks + j * D);
      ldD<D>(vv, Vs + j * D);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s = fmaf(q[d], kk[d], s); dp = fmaf(g[d], vv[d], dp); }
      const float p = expf(s - m);
      l += p;
      Dacc = fmaf(p, dp, Dacc);
    }
    const float inv = 1.f / l;
    const float Di = Dacc * inv;  // sum_j P_ij dP_ij
    float dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dq[d] = 0.f;
    for (int j = 0; j < N; ++j) {
      float kk[D], vv[D];
      ldD<D>(kk, Ks + j * D);
      ldD<D>(vv, Vs + j * D);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s = fmaf(q[d], kk[d], s); dp = fmaf(g[d], vv[d], dp); }
      const float ds = expf(s - m) * inv * (dp - Di);
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
    }
    mx[i] = m; isum[i] = inv; Dv[i] = Di;
    float *dst = dqkv + ((size_t)b * N + i) * 384 + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4 *>(dst + d) = make_float4(dq[d] * scale, dq[d + 1] * scale,
                                                         dq[d + 2] * scale, dq[d + 3] * scale);
  }
  __syncthreads();
  // ---- pass B: lane = j
  for (int j = lane; j < N; j += 64) {
    float k[D], v[D], dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { k[d] = Ks[j * D + d]; v[d] = Vs[j * D + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < N; ++i) {
      float qq[D], gg[D];
      ldD<D>(qq, Qs + i * D);
      ldD<D>(gg, Gs + i * D);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s = fmaf(qq[d] * scale, k[d], s); dp = fmaf(gg[d], v[d], dp); }
      const float p = expf(s - mx[i]) * isum[i];
      const float ds = p * (dp - Dv[i]);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dv[d] = fmaf(p, gg[d], dv[d]);
        dk[d] = fmaf(ds, qq[d] * scale, dk[d]);
      }
    }
    float *dst = dqkv + ((size_t)b * N + j) * 384 + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      *reinterpret_cast<float4 *>(dst + 128 + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
      *reinterpret_cast<float4 *>(dst + 256 + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
    }
  }
}

// The same backward on the matrix cores for N <= 48 (three 16-node tiles): one wave per (graph,
// head), every operand in registers, no LDS.  Lane (c, g) = (lane & 15, lane >> 4); "row" fragments
// X[16 t + c][4 g .. 4 g + 3] (one 16-byte load; an MFMA k-step s hands group g the head column
// 4 g + s on both sides) and "col" fragments X[16 t + 4 g + i][c].
//   lane = query:  S^T(tj,ti) = K Q^T and dP^T = V dO^T put query i = c on the lane with keys
//                  16 tj + 4 g + r in its registers: row max, sum and D_i are in-lane + two xor steps,
//                  and ds^T is the B operand of dQ^T = K(col) ds^T  (16-byte stores);
//   lane = key:    the same registers with the operands swapped give S and dP with key j = c on the
//                  lane and queries 16 ti + 4 g + r in the registers (their statistics come over by
//                  ds_bpermute), the B operands of dK^T = Q(col) ds and dV^T = dO(col) P.
// 252 MFMAs per (graph, head) at N = 40 against ~3 N^2 x 50 VALU operations on 40 of 64 lanes:
// 247 -> 70 us per layer at 2048 x 40.
typedef float ab_f4 __attribute__((ext_vector_type(4)));
#define AB_MFMA4(D, A, Bv)                                                \
  D = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (Bv).x, D, 0, 0, 0);    \
  D = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (Bv).y, D, 0, 0, 0);    \
  D = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (Bv).z, D, 0, 0, 0);    \
  D = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (Bv).w, D, 0, 0, 0)

template <int NT>
__global__ __launch_bounds__(256, 2) void encoder_attention_bwd_mfma_kernel(
    const float *__restrict__ qkv, const float *__restrict__ dO, float *__restrict__ dqkv, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, h = blockIdx.y * 4 + wave;
  const float *base = qkv + (size_t)b * N * 384 + h * 16;
  const float *gbase = dO + (size_t)b * N * 128 + h * 16;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 QR[NT], KR[NT], VR[NT], GR[NT];
  float KC[NT][4], QC[NT][4], GC[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = 16 * t + c;
    const bool on = n < N;
    const float *r = base + (size_t)(on ? n : 0) * 384 + 4 * g;
    const float4 q4 = on ? *reinterpret_cast<const float4 *>(r) : zero4;
    QR[t] = make_float4(q4.x * 0.25f, q4.y * 0.25f, q4.z * 0.25f, q4.w * 0.25f);   // 1/sqrt(16)
    KR[t] = on ? *reinterpret_cast<const float4 *>(r + 128) : zero4;
    VR[t] = on ? *reinterpret_cast<const float4 *>(r + 256) : zero4;
    GR[t] = on ? *reinterpret_cast<const float4 *>(gbase + (size_t)n * 128 + 4 * g) : zero4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * t + 4 * g + i;
      const bool om = m < N;
      const float *rc = base + (size_t)(om ? m : 0) * 384 + c;
      QC[t][i] = om ? rc[0] * 0.25f : 0.f;
      KC[t][i] = om ? rc[128] : 0.f;
      GC[t][i] = om ? gbase[(size_t)m * 128 + c] : 0.f;
    }
  }
  // ---- lane = query i = 16 ti + c ------------------------------------------------------------
  float st_m[NT], st_inv[NT], st_D[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    ab_f4 p[NT], dp[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      ab_f4 d = {0.f, 0.f, 0.f, 0.f};
      AB_MFMA4(d, KR[tj], QR[ti]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * tj + 4 * g + r >= N) d[r] = -INFINITY;
        mx = fmaxf(mx, d[r]);
      }
      p[tj] = d;
      ab_f4 e = {0.f, 0.f, 0.f, 0.f};
      AB_MFMA4(e, VR[tj], GR[ti]);
      dp[tj] = e;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f, Dacc = 0.f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(p[tj][r] - mx);   // 0 beyond N
        p[tj][r] = e;
        l += e;
        Dacc = fmaf(e, dp[tj][r], Dacc);
      }
    l += __shfl_xor(l, 16, 64);       Dacc += __shfl_xor(Dacc, 16, 64);
    l += __shfl_xor(l, 32, 64);       Dacc += __shfl_xor(Dacc, 32, 64);
    const float inv = 1.f / l, Di = Dacc * inv;   // sum_j P_ij dP_ij
    st_m[ti] = mx; st_inv[ti] = inv; st_D[ti] = Di;
    ab_f4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = p[tj][r] * inv * (dp[tj][r] - Di);
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(KC[tj][r], ds, dq, 0, 0, 0);
      }
    const int i = 16 * ti + c;
    if (i < N)
      *reinterpret_cast<float4 *>(dqkv + ((size_t)b * N + i) * 384 + h * 16 + 4 * g) =
          make_float4(dq[0] * 0.25f, dq[1] * 0.25f, dq[2] * 0.25f, dq[3] * 0.25f);
  }
  // ---- lane = key j = 16 tj + c; statistics of query 16 ti + 4 g + r from lane 4 g + r ----------
  float s2m[NT][4], s2i[NT][4], s2D[NT][4];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s2m[ti][r] = __shfl(st_m[ti], 4 * g + r, 64);
      s2i[ti][r] = __shfl(st_inv[ti], 4 * g + r, 64);
      s2D[ti][r] = __shfl(st_D[ti], 4 * g + r, 64);
    }
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    const int j = 16 * tj + c;
    ab_f4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      ab_f4 sv = {0.f, 0.f, 0.f, 0.f}, dpv = {0.f, 0.f, 0.f, 0.f};
      AB_MFMA4(sv, QR[ti], KR[tj]);
      AB_MFMA4(dpv, GR[ti], VR[tj]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = (16 * ti + 4 * g + r < N) && (j < N);
        const float pv = on ? expf(sv[r] - s2m[ti][r]) * s2i[ti][r] : 0.f;
        const float ds = pv * (dpv[r] - s2D[ti][r]);
        dv = __builtin_amdgcn_mfma_f32_16x16x4f32(GC[ti][r], pv, dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(QC[ti][r], ds, dk, 0, 0, 0);
      }
    }
    if (j < N) {
      float *dst = dqkv + ((size_t)b * N + j) * 384 + h * 16 + 4 * g;
      *reinterpret_cast<float4 *>(dst + 128) = make_float4(dk[0], dk[1], dk[2], dk[3]);
      *reinterpret_cast<float4 *>(dst + 256) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
  }
}

template <int D, int W>
static int launch_attention_bwd_valu(const float *qkv, const float *dO, float *dqkv, int B, int N,
                                     hipStream_t st) {
  const size_t lds = (size_t)W * ((N * (4 * D + 3) + 3) & ~3) * sizeof(float);
  VRP_REQUIRE(lds <= 160 * 1024, "attention_bwd: N=%d too large for head width %d", N, D);
  static VrpAttrOnce attr_set;
  if (!attr_set.done() && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_attention_bwd_kernel<D, W>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      vrp_set_error("attention_bwd: cannot raise dynamic LDS");
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL((encoder_attention_bwd_kernel<D, W>), dim3(B, (128 / D) / W), dim3(64 * W), lds, st,
                     qkv, dO, dqkv, N);
  VRP_CHECK_LAUNCH("encoder_attention_bwd");
  return 0;
}

// heads: 8 (the reference's default: matrix-core kernel for N <= 48, VALU above), 4 or 16 (VALU;
// four heads of width 32 run two waves per workgroup: Q, K, V, dO of a wave are 131 N floats of LDS)
int vrp_launch_attention_bwd(const float *qkv, const float *dO, float *dqkv, int B, int N,
                             hipStream_t st, int heads) {
  if (heads == 4) return launch_attention_bwd_valu<32, 2>(qkv, dO, dqkv, B, N, st);
  if (heads == 16) return launch_attention_bwd_valu<8, 4>(qkv, dO, dqkv, B, N, st);
  static const bool valu = getenv("VRP_ATTN_BWD_VALU") != nullptr;   // A/B aid
  if (N <= 48 && !valu) {
    if (N <= 16)
      hipLaunchKernelGGL(encoder_attention_bwd_mfma_kernel<1>, dim3(B, 2), dim3(256), 0, st, qkv, dO, dqkv, N);
    else if (N <= 32)
      hipLaunchKernelGGL(encoder_attention_bwd_mfma_kernel<2>, dim3(B, 2), dim3(256), 0, st, qkv, dO, dqkv, N);
    else
      hipLaunchKernelGGL(encoder_attention_bwd_mfma_kernel<3>, dim3(B, 2), dim3(256), 0, st, qkv, dO, dqkv, N);
    VRP_CHECK_LAUNCH("encoder_attention_bwd_mfma");
    return 0;
  }
  return launch_attention_bwd_valu<16, 4>(qkv, dO, dqkv, B, N, st);
}

extern "C" int vrp_attention_bwd(const float *qkv, const float *dO, float *dqkv, int B, int N,
                                 void *stream) {
  return vrp_launch_attention_bwd(qkv, dO, dqkv, B, N, (hipStream_t)stream, 8);
}
