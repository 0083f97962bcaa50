// Full-frame ("reference semantics", tile=None) mode of the DCT-SVD watermark
// on gfx950: ONE dense SVD per plane instead of one per 8x8 tile
// (app_dct_svd_single.py:172-177 embed, :205-218 extract, :297-301 detect).
//
// Algorithm:
//  * the orthonormal DCT cancels (svd(D_H Y D_W^T) = (D_H Ux) S (D_W Vx)^T), so
//    embed/sigma/detect work on the pixel plane itself;
//  * block one-sided Jacobi on the SHORT side: A (L x M, L <= M; the plane or
//    its transpose) is augmented with the identity, Aug = [A | I_L]; row
//    rotations turn it into [B | Qt] with B = Qt A having mutually orthogonal
//    rows, |b_i| = sigma_i, Qt^T = left singular vectors.  Rows are processed in
//    blocks of RB: a step takes disjoint block pairs (round-robin tournament),
//    forms each pair's 2RB x 2RB Gram matrix (k_rf_gram), runs one cyclic
//    two-sided Jacobi sweep on it in LDS (k_rf_inner -> rotation block R) and
//    applies R^T to the 2RB rows of Aug as a small GEMM (k_rf_apply);
//  * embed is V-free: Yw = Y + Q diag(alpha*sw_rank(i) / sigma_i) B.
// DCT-domain factors (watermark side / extract) use plain tiled SGEMMs with the
// DCT basis matrices.
#include <chrono>
#include <thread>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <new>
#include <numeric>
#include <utility>
#include <vector>

#include "wm_internal.h"

using namespace wmi;

namespace {

constexpr int RB = 32;        // rows per Jacobi block
constexpr int RP = 2 * RB;    // rows per block pair
#ifndef WM_GRAM_CC
#define WM_GRAM_CC 128
#endif
constexpr int GRAM_CC = WM_GRAM_CC;  // columns per Gram partial (128 or 256)
constexpr int MAX_SWEEPS = 40;
constexpr int FULL_INNER_SWEEPS = 0;   // outer sweeps whose every step runs the full 63-step inner schedule
constexpr float CONV_COS = 2e-5f;   // float32 Gram entries resolve cos down to ~eps*sqrt(M)
// sigma-only calls (extract, detect): row norms are exact to O(cos^2), and the sweep that
// observes max cos < c still rotates (leaving ~c^2), so they may stop an order earlier (2e-3 already costs 7e-5 relative on dense spectra: tools/ff_sigma_thr.py)
constexpr float CONV_COS_SIGMA = 2e-4f;
constexpr float SKIP_FRACTION = 0.25f;    // a block pair below this fraction of the (embed) stopping cosine is left alone
constexpr double RESIDUE_RHO = 10.0;      // |A0 b_i^T| / |b_i|^2 above this: b_i is not a singular direction at all
constexpr double T_SWITCH = 200.0;        // |A0 b_i^T| / |b_i| is used for s_i >= T_SWITCH * (residual cosine) * s_max
constexpr int HIER_F16_DEFAULT = 3;         // two-level scheme, split-f16 operands on the f16 matrix pipe: bit 0 Gram tiles (k_hgram_h), bit 1 rotation products (k_happly_h); WM_RF_HIER_F16 overrides
constexpr int HIER_MIN_PLANES = 20;         // ... with the f32 kernels
constexpr int HIER_MIN_PLANES_F16 = 3;      // planes per call from which the two-level scheme is the default when its split-f16 kernels apply        // planes per call from which the two-level scheme (wm_ref_hier.inc) is the default
constexpr int DEFAULT_QUEUES = 2;         // plane groups of a batched Jacobi, each on its own HIP queue
constexpr double DRIFT_TOL = 1e-2;        // |T[:, i]| / |b_i|^2 may differ from 1 by the scale drift, not more
constexpr double NULL_ROW_RATIO = 1e-5;   // rows below this fraction of |A|_F do not take part in the convergence test
constexpr double NULL_RATIO = 1e-6;   // embed: singular directions below this fraction of s_1 get no watermark energy

// ---------------------------------------------------------------------------
// generic row-major SGEMM:  C = alpha * op(A) op(B) + beta * C  on MFMA
// (v_mfma_f32_32x32x2_f32).  64x64 tile of C per workgroup, one 32x32 quadrant per wave,
// K step 32 staged through LDS in operand-friendly layouts - As[m][k] pitch 33 (the A operand
// A[m0 + lane%32][k + lane/32] hits 32 banks), Bs[k][n] pitch 65 - with the next step's global
// loads issued before this step's 16 MFMAs.  Plain-library-GEMM shaped work of the full-frame
// mode (DCT as two GEMMs, T = A0 B^T, U diag V^T products).
// ---------------------------------------------------------------------------
typedef float v16f_s __attribute__((ext_vector_type(16)));

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_sgemm(const int M, const int N, const int K, const float alpha,
                                              const float* __restrict__ A, const int lda, const size_t sA,
                                              const float* __restrict__ B, const int ldb, const size_t sB,
                                              const float beta, float* __restrict__ C, const int ldc, const size_t sC) {
  constexpr int KS = 32;
  A += (size_t)blockIdx.z * sA; B += (size_t)blockIdx.z * sB; C += (size_t)blockIdx.z * sC;   // batch: one product per grid.z
  __shared__ float As[2][64][KS + 1];   // [m][k]
  __shared__ float Bs[2][KS][64 + 1];   // [k][n]
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, j = lane & 31, h = lane >> 5;
  const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
  // staging maps: 2048 elements of each tile, 8 per thread, lanes along the contiguous global axis
  //   A stored [m][k] (TA = false): e -> m = e >> 5, k = e & 31;   A stored [k][m] (TA): k = e >> 6, m = e & 63
  //   B stored [k][n] (TB = false): e -> k = e >> 6, n = e & 63;   B stored [n][k] (TB): n = e >> 5, k = e & 31
  float ra[8], rb[8];
  auto fetch = [&](const int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = t + 256 * i;
      const int m = TA ? (e & 63) : (e >> 5), ka = TA ? (e >> 6) : (e & 31);
      const int gm = bm + m, gk = k0 + ka;
      const bool oka = gm < M && gk < K;
      const size_t ia = TA ? (size_t)(oka ? gk : 0) * lda + (oka ? gm : 0) : (size_t)(oka ? gm : 0) * lda + (oka ? gk : 0);
      const float va = A[ia];
      ra[i] = oka ? va : 0.0f;
      const int n = TB ? (e >> 5) : (e & 63), kb = TB ? (e & 31) : (e >> 6);
      const int gn = bn + n, gkb = k0 + kb;
      const bool okb = gn < N && gkb < K;
      const size_t ib = TB ? (size_t)(okb ? gn : 0) * ldb + (okb ? gkb : 0) : (size_t)(okb ? gkb : 0) * ldb + (okb ? gn : 0);
      const float vb = B[ib];
      rb[i] = okb ? vb : 0.0f;
    }
  };
  auto stash = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = t + 256 * i;
      const int m = TA ? (e & 63) : (e >> 5), ka = TA ? (e >> 6) : (e & 31);
      As[buf][m][ka] = ra[i];
      const int n = TB ? (e >> 5) : (e & 63), kb = TB ? (e & 31) : (e >> 6);
      Bs[buf][kb][n] = rb[i];
    }
  };
  const int m0 = (wv >> 1) * 32, n0 = (wv & 1) * 32;
  // Two-level accumulation: every K step of 32 is summed from zero in the matrix core (which adds its products to
  // C one by one, round-to-nearest: tools/mfma_round_probe.hip) and the step sums are added up by the VALU.  One long
  // chain loses every term that follows a dominant one below that term's ulp - the DC coefficient of a DCT plane
  // leads its row by 2^11 - which made sigma_1 of the watermark-side SVD 5e-6 (1080p) to 4e-5 (8K) low.
  v16f_s acc = {0};
  fetch(0);
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += KS, buf ^= 1) {
    stash(buf);
    if (k0 + KS < K) fetch(k0 + KS);
    __syncthreads();                    // double-buffered LDS: one barrier per K step
    const float* pa = &As[buf][m0 + j][h];
    const float* pb = &Bs[buf][h][n0 + j];
    v16f_s part = {0};
#pragma unroll
    for (int kk = 0; kk < KS / 2; ++kk)
      part = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kk], pb[2 * kk * 65], part, 0, 0, 0);
    acc += part;
  }
  const int gn = bn + n0 + j;
  if (gn < N) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int gm = bm + m0 + 8 * (v / 4) + 4 * h + (v % 4);
      if (gm < M) {
        float* c = C + (size_t)gm * ldc + gn;
        *c = (beta == 0.0f) ? alpha * acc[v] : __builtin_fmaf(beta, *c, alpha * acc[v]);
      }
    }
  }
}

// `batch` products per launch (grid.z); sA / sB / sC = elements between consecutive operands (0: shared)
int sgemm_b(wm_ctx* ctx, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, size_t sA,
            const float* B, int ldb, size_t sB, float beta, float* C, int ldc, size_t sC, int batch) {
  if (M <= 0 || N <= 0 || batch <= 0) return WM_OK;
  const dim3 grid((N + 63) / 64, (M + 63) / 64, batch), block(256);
  if (ta && tb) hipLaunchKernelGGL((k_sgemm<true, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC);
  else if (ta) hipLaunchKernelGGL((k_sgemm<true, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC);
  else if (tb) hipLaunchKernelGGL((k_sgemm<false, true>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC);
  else hipLaunchKernelGGL((k_sgemm<false, false>), grid, block, 0, ctx->stream, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

int sgemm(wm_ctx* ctx, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda,
          const float* B, int ldb, float beta, float* C, int ldc) {
  return sgemm_b(ctx, ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
}

// ---------------------------------------------------------------------------
// Aug = [A | I]:  A[i][j] = src(i, j) (or src(j, i) when `transpose`), rows
// i >= L are zero padding; the identity block is Lp x Lp.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_rf_load(const T* __restrict__ src, const size_t src_stride, const size_t src_plane_stride,
                          const int transpose, float* __restrict__ aug, const size_t aug_plane_stride,
                          const int ld, const int L, const int Lp, const int M) {
  const int i = blockIdx.y;
  src += (size_t)blockIdx.z * src_plane_stride;
  aug += (size_t)blockIdx.z * aug_plane_stride;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M + Lp; j += gridDim.x * blockDim.x) {
    float v;
    if (j < M) v = (i < L) ? (float)(transpose ? src[(size_t)j * src_stride + i] : src[(size_t)i * src_stride + j]) : 0.0f;
    else v = (j - M == i) ? 1.0f : 0.0f;
    aug[(size_t)i * ld + j] = v;
  }
}

// ---------------------------------------------------------------------------
// Gram partials: for pair p = (I, J) and column chunk ch,
//   G[r][c] = sum_{k in chunk} X[r][k] X[c][k],  X = rows of blocks I,J; stored as the three quadrants
//   partial[p][ch] = [G_II | G_IJ | G_JJ] (32 x 32 each) - G_JI is the transpose (k_rf_inner mirrors it)
// ---------------------------------------------------------------------------
// MFMA form: the 64 x 128 chunk of X is staged row-major in LDS (pitch 129: the operand reads
// X[i0 + lane%32][k + lane/32] of a wave hit 32 distinct banks twice), wave w owns the 32 x 32
// quadrant (w >> 1, w & 1) of the Gram matrix and issues 64 v_mfma_f32_32x32x2_f32 whose A and B
// operands both come from X (D = X_I X_J^T).
typedef float v16f_g __attribute__((ext_vector_type(16)));
constexpr int GP = GRAM_CC + 1;
constexpr int GRAM_PART = 3 * RB * RB;      // floats per (pair, chunk) partial: the quadrants (I,I), (I,J), (J,J)

__global__ __launch_bounds__(256) void k_rf_gram(const float* __restrict__ aug, const size_t aug_plane_stride,
                                                const int ld, const int M, const int2* __restrict__ pairs,
                                                float* __restrict__ partials) {
  __shared__ float Xs[RP][GP];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, j = lane & 31, h = lane >> 5;
  const int p = blockIdx.x, ch = blockIdx.y, nch = gridDim.y;
  aug += (size_t)blockIdx.z * aug_plane_stride;
  partials += (size_t)blockIdx.z * gridDim.x * nch * GRAM_PART;
  const int2 pr = pairs[p];
  const int c_begin = ch * GRAM_CC;
  {
    // 64 rows x GRAM_CC columns: thread t takes column t % GRAM_CC of rows t / GRAM_CC + RSTEP i; all loads are
    // issued before the LDS stores
    constexpr int RSTEP = 256 / GRAM_CC, NLD = RP / RSTEP;
    const int c = t % GRAM_CC, r0 = t / GRAM_CC, gc = c_begin + c;
    const bool valid = gc < M;
    const int gcc = valid ? gc : M - 1;
    float v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int r = r0 + RSTEP * i;
      const int grow = (r < RB) ? pr.x * RB + r : pr.y * RB + (r - RB);
      v[i] = aug[(size_t)grow * ld + gcc];
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) Xs[r0 + RSTEP * i][c] = valid ? v[i] : 0.0f;
  }
  __syncthreads();
  // G is symmetric: the quadrant (J, I) is the transpose of (I, J) and is neither computed nor stored - a partial
  // is [Q_II | Q_IJ | Q_JJ], 3 x 32 x 32 floats (GRAM_PART); wave 2 has nothing to do after the staging
  if (wv == 2) return;
  const int i0 = (wv >> 1) * 32, j0 = (wv & 1) * 32;
  const float* ra = &Xs[i0 + j][h];
  const float* rb = &Xs[j0 + j][h];
  v16f_g acc = {0};
#pragma unroll 16
  for (int kk = 0; kk < GRAM_CC / 2; ++kk)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * kk], rb[2 * kk], acc, 0, 0, 0);
  float* out = partials + ((size_t)p * nch + ch) * GRAM_PART + (wv == 3 ? 2 : wv) * (RB * RB);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int row = 8 * (v / 4) + 4 * h + (v % 4);
    out[row * RB + j] = acc[v];
  }
}

// ---------------------------------------------------------------------------
// One cyclic two-sided Jacobi sweep on the pair's 64x64 Gram matrix in LDS.
// Parallel (round-robin) ordering: 63 steps x 32 disjoint rotations; de Rijk
// rule per rotation (larger diagonal entry to the lower index).  Outputs the
// accumulated rotation block R (G' = R^T G R) and the sweep-wide maximum of
// |G_rc| / sqrt(G_rr G_cc) (convergence measure, before rotating).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rr_elem(const int pos, const int step) {
  // round-robin tournament over 64 players: position 0 is fixed, the others rotate
  if (pos == 0) return 0;
  int v = pos - 1 - step;          // step < 63
  if (v < 0) v += RP - 1;
  return v + 1;
}

// cross_only != 0: only the 32 x 32 pairs between the two row blocks are rotated
// (bipartite schedule, 32 steps); the within-block pairs are covered once per
// outer sweep by the full 63-step schedule (every block sits in exactly one
// pair of the sweep's first step).
// Thread layout: thread (kr = t >> 5, k2 = t & 31) owns column pair k2 and the row pairs kr + KR i: with
// INNER_NT = 512 threads two 2x2 blocks of G and four (row, column pair) items of R per step.
// What bounds a step (measured, profiles/r02_fullframe_inner.md): in-kernel stamps give 1 716 cycles per
// cross-only step for 1024 AND for 512 threads (2 540 with the full 63-step schedule, whose index arithmetic
// branches) - a step is the LDS pipe's time, not the instruction count: every step reads and rewrites all of
// G and R (19.5 k dword accesses; 8 192 of them ds_write_b32 at 64 B/clk/CU = 512 cycles, the reads 350) in
// lock-step bursts between two barriers, plus the angle arithmetic's two v_rsq_f32 chains (~250 cycles)
// that nothing else can run under.  256 threads (one wave per SIMD) lose the latency hiding (39.7 us per
// solve against 30.1); 512 threads issue the per-pair angle and index arithmetic half as often as 1024 and
// are 14 % faster on an 8-plane batch (98 against 86 frames/s), equal on a single plane.
// Two-level (super-block) scheme, see "hierarchical block Jacobi" further down: a super-pair is at most HSB
// 32-row blocks (HN rows); its Gram matrix G_s and accumulated rotation Q_s are HN x HN arrays (pitch HN) in
// global memory, and a stage rotates HU disjoint block pairs ("units") of it.
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int HSB = 12;
constexpr int HN = HSB * RB;
constexpr int HU = HSB / 2;
#ifndef WM_INNER_NT
#define WM_INNER_NT 512
#endif
constexpr int INNER_NT = WM_INNER_NT;
constexpr int INNER_NW = INNER_NT / 64;
constexpr int INNER_KR = INNER_NT / 32;        // row pairs a thread column covers per pass (stride between a thread's row pairs)
constexpr int INNER_NB = 32 / INNER_KR;        // 2x2 blocks of G per thread

__global__ __launch_bounds__(INNER_NT) void k_rf_inner(const float* __restrict__ partials, const int nch,
                                                      float* __restrict__ Rout, unsigned* __restrict__ maxcos_bits,
                                                      const float* __restrict__ floor2, const int cross_only,
                                                      int* __restrict__ skip_flags, const float skip_thr,
                                                      const int* __restrict__ h_units, float* __restrict__ h_Gs,
                                                      int* __restrict__ h_anyrot, const int h_nsp,
                                                      float* __restrict__ h_Rpk, int* __restrict__ h_skipT, const int h_f16) {
  // h_units != NULL: two-level scheme.  blockIdx.x = super-pair * HU + unit; the unit's two 32-row blocks (local
  // indices la, lb of the super-pair) are read straight out of the tracked Gram matrix G_s (no partial sums), and
  // the rotated 64 x 64 matrix R^T G R is written back into G_s at the end.
  __shared__ float GG[2][RP][RP + 1];   // double-buffered: a step reads one copy, writes the other
  float (*G)[RP + 1] = GG[0];
  __shared__ float R[RP][RP + 1];
  __shared__ float red[INNER_NW];
  const int t = threadIdx.x, p = blockIdx.x;
#if defined(WM_POISON_LDS)     // diagnostic: every word of the kernel's LDS starts as a NaN - a read of a word the kernel never wrote shows in the results
  for (int i = threadIdx.x; i < 2 * RP * (RP + 1); i += INNER_NT) (&GG[0][0][0])[i] = __int_as_float(0x7fc00000);
  for (int i = threadIdx.x; i < RP * (RP + 1); i += INNER_NT) (&R[0][0])[i] = __int_as_float(0x7fc00000);
  if (threadIdx.x < INNER_NW) red[threadIdx.x] = __int_as_float(0x7fc00000);
  __syncthreads();
#endif
#if defined(WM_INNER_DIAG)     // diagnostic build only (tools/): where one inner solve spends its cycles
  unsigned long long st0 = __builtin_amdgcn_s_memtime(), st1 = 0, st2 = 0, st3 = 0;
#endif
  partials += (size_t)blockIdx.z * gridDim.x * nch * GRAM_PART;
  Rout += (size_t)blockIdx.z * gridDim.x * RP * RP;
  maxcos_bits += blockIdx.z;
  // rows whose squared norm is below this plane's floor are numerically null (rounding residue of
  // a rank-deficient plane): their mutual cosines are O(1) noise and must not hold convergence up
  const float fl2 = floor2[blockIdx.z];
  int la = 0, lb = 0;
  float* gs = nullptr;
  if (h_units) {
    const int unit = h_units[p];
    if (unit < 0 || !((unit >> 16) & 1)) return;   // no such unit in this super-pair at this stage / an idle pair
    la = unit & 0xff; lb = (unit >> 8) & 0xff;
    gs = h_Gs + ((size_t)blockIdx.z * h_nsp + p / HU) * HN * HN;
  }
  const float* src = partials + (size_t)p * nch * GRAM_PART;
  // Sum of the column-chunk partials, as 16-byte loads with up to 16 chunks (128 VGPRs) in flight per thread: the
  // workgroup is alone on its CU and this phase is one CU's fetch rate (profiles/r02_fullframe_inner.md).
  // A partial holds the quadrants (I,I), (I,J), (J,J) (k_rf_gram): float4 f = t + INNER_NT i < 768 is quadrant f >> 8,
  // row (f & 255) >> 3, columns 4 (f & 7) ..; the chunks are added in index order.
  constexpr int NF4 = GRAM_PART / 4;
  constexpr int PER4 = (NF4 + INNER_NT - 1) / INNER_NT;
  constexpr int UN = 16;
  float4 acc[PER4];
#pragma unroll
  for (int i = 0; i < PER4; ++i) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (h_units) {
    for (int f = t; f < RP * RP / 4; f += INNER_NT) {
      const int r = f >> 4, c4 = (f & 15) * 4;
      const int gr = (r < RB) ? la * RB + r : lb * RB + r - RB;
      const int gc = (c4 < RB) ? la * RB + c4 : lb * RB + c4 - RB;
      const float4 v = *reinterpret_cast<const float4*>(gs + (size_t)gr * HN + gc);
      G[r][c4] = v.x; G[r][c4 + 1] = v.y; G[r][c4 + 2] = v.z; G[r][c4 + 3] = v.w;
      if (!cross_only) {
#pragma unroll
        for (int k = 0; k < 4; ++k) R[r][c4 + k] = (r == c4 + k) ? 1.0f : 0.0f;
      }
    }
  } else {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int ch = 0; ch < nch; ch += UN) {
      float4 v[UN][PER4];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int c = min(ch + u, nch - 1);            // past the end: a repeated (cached) load, masked below
#pragma unroll
        for (int i = 0; i < PER4; ++i) v[u][i] = src4[(size_t)c * NF4 + min(t + INNER_NT * i, NF4 - 1)];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const bool in = ch + u < nch;
#pragma unroll
        for (int i = 0; i < PER4; ++i) {
          acc[i].x += in ? v[u][i].x : 0.0f; acc[i].y += in ? v[u][i].y : 0.0f;
          acc[i].z += in ? v[u][i].z : 0.0f; acc[i].w += in ? v[u][i].w : 0.0f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
      const int f = t + INNER_NT * i;
      if (f < NF4) {
        const int qd = f >> 8, r = ((f & 255) >> 3) + (qd == 2 ? RB : 0), c = 4 * (f & 7) + (qd >= 1 ? RB : 0);
        const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          G[r][c + k] = av[k];
          if (qd == 1) G[c + k][r] = av[k];           // the (J, I) quadrant is the transpose
          if (!cross_only) {                          // the cross-only path keeps R in registers
            R[r][c + k] = (r == c + k) ? 1.0f : 0.0f;
            if (qd == 1) R[c + k][r] = 0.0f;
          }
        }
      }
    }
  }
  __syncthreads();
#if defined(WM_INNER_DIAG)
  st1 = __builtin_amdgcn_s_memtime();
#endif
  // largest cosine between two rows of the pair, from the sums still in registers and the diagonal in LDS
  float mx = 0.0f;
  if (h_units) {
    for (int f = t; f < RP * RP; f += INNER_NT) {
      const int r = f >> 6, c = f & 63;
      if (r < c) {
        const float grr = G[r][r], gcc = G[c][c];
        if (grr > fl2 && gcc > fl2) mx = fmaxf(mx, fabsf(G[r][c]) * __builtin_amdgcn_rsqf(grr * gcc));
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < PER4; ++i) {
    const int f = t + INNER_NT * i;
    if (f < NF4) {
      const int qd = f >> 8, r = ((f & 255) >> 3) + (qd == 2 ? RB : 0), c = 4 * (f & 7) + (qd >= 1 ? RB : 0);
      const float grr = G[r][r];
      const float av[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gcc = G[c + k][c + k];
        if (r != c + k && grr > fl2 && gcc > fl2) mx = fmaxf(mx, fabsf(av[k]) * __builtin_amdgcn_rsqf(grr * gcc));
      }
    }
  }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  __shared__ int s_skip;
  if (t == 0) {
    float m = red[0];
#pragma unroll
    for (int i = 1; i < INNER_NW; ++i) m = fmaxf(m, red[i]);
    atomicMax(maxcos_bits, __float_as_uint(m));
    // a pair whose Gram matrix is already diagonal far below the stopping cosine has nothing left to
    // rotate (the sweeps converge quadratically, so this is the whole last sweep): no solve, and the
    // apply kernel skips the pair's rows as well
    s_skip = (m < skip_thr) ? 1 : 0;
    skip_flags[(size_t)blockIdx.z * gridDim.x + p] = s_skip;
    if (h_skipT) h_skipT[(size_t)blockIdx.z * gridDim.x + p] = s_skip;       // this stage's copy, read by k_happly at the end of the super-step
  }
  __syncthreads();
  if (s_skip) return;
  if (h_units && t == 0) h_anyrot[(size_t)blockIdx.z * h_nsp + p / HU] = 1;     // this super-pair's rows do get rotated
  // two-level scheme: R also in the order k_happly's waves consume it - for the wave that owns output rows 32 hf .. of the
  // pair, lane (j, h), MFMA step kk: R[2 kk + h][32 hf + j] at ((hf * 8 + kk / 4) * 64 + j + 32 h) * 4 + kk % 4, so that four
  // steps' operands are one 16-byte load and a wave's load is 1 KB contiguous
  float* rpk = h_Rpk ? h_Rpk + ((size_t)blockIdx.z * gridDim.x + p) * RP * RP : nullptr;
  auto pk_index = [](const int r, const int c) { return (((c >> 5) * 8 + (r >> 3)) * 64 + (c & 31) + 32 * (r & 1)) * 4 + ((r >> 1) & 3); };
  // h_f16: the same 16 KB as split f16 for k_happly_h (v_mfma_f32_32x32x16_f16: lane (i, kg) holds 8 consecutive k):
  // R[k][c] = hi + lo' / 2048 with hi = f16(R) (0 below f16's normal range), lo' = f16((R - hi) * 2048);
  // half index ((c / 32 * 4 + k / 16) * 64 + c % 32 + 32 * (k / 8 % 2)) * 8 + k % 8, hi in the first 4096 halfs, lo' behind
  auto put_r = [&](const int r, const int c, const float v) {
    if (!h_f16) { rpk[pk_index(r, c)] = v; return; }
    _Float16* hp = reinterpret_cast<_Float16*>(rpk);
    const int idx = (((c >> 5) * 4 + (r >> 4)) * 64 + (c & 31) + 32 * ((r >> 3) & 1)) * 8 + (r & 7);
    const _Float16 hi = fabsf(v) < 6.2e-5f ? (_Float16)0.0f : (_Float16)v;
    hp[idx] = hi;
    hp[RP * RP + idx] = (_Float16)((v - (float)hi) * 2048.0f);
  };
  // two-level scheme: the rotated Gram matrix (pitch RP + 1 in LDS) goes back into G_s
  auto writeback = [&](const float* gf) {
    for (int f = t; f < RP * RP / 4; f += INNER_NT) {
      const int r = f >> 4, c4 = (f & 15) * 4;
      const int gr = (r < RB) ? la * RB + r : lb * RB + r - RB;
      const int gc = (c4 < RB) ? la * RB + c4 : lb * RB + c4 - RB;
      const float* q = gf + r * (RP + 1) + c4;
      *reinterpret_cast<float4*>(gs + (size_t)gr * HN + gc) = make_float4(q[0], q[1], q[2], q[3]);
    }
  };

  // Every lane computes the rotation of pair (lane & 31) - the wave's two halves redundantly - so the
  // rotation of the thread's COLUMN pair k2 is in its own registers, and those of its ROW pairs
  // kr + KR i (kr uniform per half-wave) come from lanes kr + KR i by v_readlane: no rotation table in
  // LDS, one workgroup barrier per step.
  const int k2 = t & 31;
  const int wv_s = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool hi = (t & 32) != 0;
  const int kr = 2 * wv_s + (hi ? 1 : 0);
  const int n_inner = cross_only ? RB : RP - 1;
#if defined(WM_INNER_DIAG)
  st2 = __builtin_amdgcn_s_memtime();
#endif
  float* out = Rout + (size_t)p * RP * RP;
#if defined(WM_INNER_R_LDS)      // A/B only: R through LDS on every step (the round-2a kernel)
  if (false) {
#else
  if (cross_only) {
#endif
    // Cross-block schedule (32 steps; every step of a sweep but the first): block-I row k meets block-J row
    // (k + step) mod 32.  R never touches LDS here: thread (kr, k2) keeps R[r][k2] (its block-I column, fixed) and
    // R[r][32 + (k2 + step) mod 32] (the block-J column of its current pair) for its rows r = kr + KR j in
    // registers; after a step the block-J values move one lane down (ds_bpermute, no LDS memory), so that every
    // lane holds the column of its next partner.  After 32 steps they are back where they started.  This takes
    // the 8 192 R accesses (half of them ds_write_b32 at 64 B/clk) out of each step's LDS time.
    float ri[2 * INNER_NB], rj[2 * INNER_NB];
#pragma unroll
    for (int j = 0; j < 2 * INNER_NB; ++j) {
      const int r = kr + INNER_KR * j;
      ri[j] = (r == k2) ? 1.0f : 0.0f;            // R starts as the identity (the LDS copy is not used on this path)
      rj[j] = (r == RB + k2) ? 1.0f : 0.0f;
    }
    const int shl = ((t & 32) | ((t + 1) & 31)) << 2;     // byte address of the lane one up within the half-wave
    // The LDS offsets of the moving (block-J) row and column are carried from step to step (add + wrap) instead of
    // rebuilt from (k + step) mod 32, and the two G buffers are compile-time bases (two steps per loop iteration): the
    // address arithmetic in front of a step's reads goes from ~35 to ~12 instructions (reads phase 490 -> 360 cycles).
    // A row pair's rotation comes by v_readlane from the lane that computed it; fetching it by ds_bpermute instead
    // saves 16 more instructions and is SLOWER (its latency sits in the step's dependent chain; 8 planes 107 -> 101).
    constexpr int PITCH = RP + 1;
    float* const gbuf0 = &GG[0][0][0];
    float* const gbuf1 = &GG[1][0][0];
    const int dp = k2 * (PITCH + 1);                       // G[p2][p2]
    int cq = RB + k2;                                      // column q2 of this step's pair
    int dq = cq * (PITCH + 1);                             // G[q2][q2]
    int rp_[INNER_NB], rq_[INNER_NB];
#pragma unroll
    for (int i = 0; i < INNER_NB; ++i) {
      rp_[i] = (kr + INNER_KR * i) * PITCH;                // row p1 (fixed)
      rq_[i] = (RB + kr + INNER_KR * i) * PITCH;           // row q1 (moves one row down per step, wraps to row 32)
    }
#if defined(WM_INNER_DIAG)
    int diag_step = 0;
#endif
    auto one_step = [&](const float* __restrict__ Gs, float* __restrict__ Gd) {
#if defined(WM_INNER_DIAG)
      const unsigned long long d0 = __builtin_amdgcn_s_memtime();
#endif
      const float app = Gs[dp], aqq = Gs[dq], apq = Gs[k2 * PITCH + cq];
      float g[INNER_NB][4];
#pragma unroll
      for (int i = 0; i < INNER_NB; ++i) {
        g[i][0] = Gs[rp_[i] + k2]; g[i][1] = Gs[rp_[i] + cq]; g[i][2] = Gs[rq_[i] + k2]; g[i][3] = Gs[rq_[i] + cq];
      }
#if defined(WM_INNER_DIAG)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long d1 = __builtin_amdgcn_s_memtime();
#endif
      const float tau = aqq - app, g2 = apq + apq;
      const float ta = fabsf(tau) + 1e-18f;
      const float ih = __builtin_amdgcn_rsqf(fmaf(g2, g2, ta * ta));
      const float x = fmaf(0.5f * ta, ih, 0.5f);
      const float rx = __builtin_amdgcn_rsqf(x);
      float c0 = x * rx, s0 = (apq * ih) * rx;
      const float nrm = fmaf(-0.5f, fmaf(c0, c0, s0 * s0), 1.5f);
      c0 *= nrm; s0 *= nrm;
      const bool sw = tau > 0.0f;
      const float C2 = sw ? s0 : c0, S2 = sw ? -c0 : -s0;
      const f2v cs2 = {C2, S2}, ns2 = {-S2, C2};
#pragma unroll
      for (int i = 0; i < INNER_NB; ++i) {
        const int la = 2 * wv_s + INNER_KR * i;            // the lanes that hold the row pairs' rotations (wave-uniform)
        const float C1a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C2), la));
        const float C1b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C2), la + 1));
        const float S1a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S2), la));
        const float S1b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S2), la + 1));
        const float C1 = hi ? C1b : C1a, S1 = hi ? S1b : S1a;
        // two-wide (v_pk_mul_f32 / v_pk_fma_f32): X = row p1 and Y = row q1 with the columns rotated, then the rows
        const f2v X = g[i][0] * cs2 + g[i][1] * ns2;           // (C2 gpp - S2 gpq, S2 gpp + C2 gpq)
        const f2v Y = g[i][2] * cs2 + g[i][3] * ns2;
        const f2v P = C1 * X - S1 * Y, Q = S1 * X + C1 * Y;
        Gd[rp_[i] + k2] = P.x; Gd[rp_[i] + cq] = P.y;
        Gd[rq_[i] + k2] = Q.x; Gd[rq_[i] + cq] = Q.y;
      }
#pragma unroll
      for (int j = 0; j < 2 * INNER_NB; ++j) {
        const f2v rr = ri[j] * cs2 + rj[j] * ns2;              // (C2 rp - S2 rq, S2 rp + C2 rq)
        ri[j] = rr.x;
        rj[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(shl, __float_as_int(rr.y)));   // lane k2 takes lane k2 + 1's block-J column
      }
      // next step: column q2 and the rows q1 move on by one, from 63 back to 32
      const bool wc = cq == RP - 1;
      cq = wc ? RB : cq + 1;
      dq = wc ? RB * (PITCH + 1) : dq + (PITCH + 1);
#pragma unroll
      for (int i = 0; i < INNER_NB; ++i) rq_[i] = (rq_[i] == (RP - 1) * PITCH) ? RB * PITCH : rq_[i] + PITCH;
#if defined(WM_INNER_DIAG)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long d2 = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#if defined(WM_INNER_DIAG)
      if (diag_step++ == 7 && t == 0 && p == 1 && blockIdx.z == 0) {
        const unsigned long long d3 = __builtin_amdgcn_s_memtime();
        printf("inner step diag: reads %llu  angle+update+writes+bpermute %llu  barrier %llu cycles\n", d1 - d0, d2 - d1, d3 - d2);
      }
#endif
    };
    for (int it = 0; it < RB / 2; ++it) {      // G starts in buffer 0 and is back there after an even number of steps
      one_step(gbuf0, gbuf1);
      one_step(gbuf1, gbuf0);
    }
#if defined(WM_INNER_DIAG)
    st3 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < 2 * INNER_NB; ++j) {
      const int r = kr + INNER_KR * j;
      // (two-level scheme: Rout in the packed order as well - k_hupdate takes its MFMA operands straight from it)
      out[h_units ? pk_index(r, k2) : r * RP + k2] = ri[j];
      out[h_units ? pk_index(r, RB + k2) : r * RP + RB + k2] = rj[j];
      if (rpk) { put_r(r, k2, ri[j]); put_r(r, RB + k2, rj[j]); }
    }
    if (h_units) writeback(gbuf0);             // 32 steps: the result is back in buffer 0, behind the last step's barrier
#if defined(WM_INNER_DIAG)
    __syncthreads();
    if (t == 0 && p == 1 && blockIdx.z == 0) {
      const unsigned long long st4 = __builtin_amdgcn_s_memtime();
      printf("inner diag (cross, R in registers): load %llu maxcos %llu loop %llu (%llu per step) store %llu cycles\n", st1 - st0,
             st2 - st1, st3 - st2, (st3 - st2) / 32ull, st4 - st3);
    }
#endif
    return;
  }
  for (int step = 0; step < n_inner; ++step) {
    float (*Gn)[RP + 1] = GG[(step + 1) & 1];
    // ---- indices of this step's pairs (cross_only: block-I row k with block-J row (k + step) mod 32) ----
    const int a2 = cross_only ? k2 : rr_elem(k2, step);
    const int b2 = cross_only ? RB + ((k2 + step) & (RB - 1)) : rr_elem(RP - 1 - k2, step);
    const int p2 = min(a2, b2), q2 = max(a2, b2);
    int p1[INNER_NB], q1[INNER_NB];
#pragma unroll
    for (int i = 0; i < INNER_NB; ++i) {
      const int k1 = kr + INNER_KR * i;
      const int a1 = cross_only ? k1 : rr_elem(k1, step);
      const int b1 = cross_only ? RB + ((k1 + step) & (RB - 1)) : rr_elem(RP - 1 - k1, step);
      p1[i] = min(a1, b1); q1[i] = max(a1, b1);
    }
    // ---- every LDS read of the step, issued before the angle arithmetic ----
    const float app = G[p2][p2], aqq = G[q2][q2], apq = G[p2][q2];
    float g[INNER_NB][4], rp_[2 * INNER_NB], rq_[2 * INNER_NB];
#pragma unroll
    for (int i = 0; i < INNER_NB; ++i) {
      g[i][0] = G[p1[i]][p2]; g[i][1] = G[p1[i]][q2]; g[i][2] = G[q1[i]][p2]; g[i][3] = G[q1[i]][q2];
    }
#pragma unroll
    for (int j = 0; j < 2 * INNER_NB; ++j) {
      const int r = kr + INNER_KR * j;
      rp_[j] = R[r][p2]; rq_[j] = R[r][q2];
    }
    // ---- rotation of column pair k2 ----
    const float tau = aqq - app, g2 = apq + apq;
    const float ta = fabsf(tau) + 1e-18f;   // all-zero (padding) rows: cos = 1, sin = 0
    const float ih = __builtin_amdgcn_rsqf(fmaf(g2, g2, ta * ta));   // 1/h, h^2 = tau^2 + 4 apq^2
    const float x = fmaf(0.5f * ta, ih, 0.5f);                       // cos^2 in [0.5, 1]
    const float rx = __builtin_amdgcn_rsqf(x);
    float c0 = x * rx, s0 = (apq * ih) * rx;                         // cos, sin * sign(apq)
    // v_rsq_f32 leaves cos^2 + sin^2 a few 1e-8 off 1, always to the same side: after ~3e4 rotations
    // the rows' scale has drifted by 1e-4.  One Newton step on the pair's norm removes the bias.
    const float nrm = fmaf(-0.5f, fmaf(c0, c0, s0 * s0), 1.5f);
    c0 *= nrm; s0 *= nrm;
    const bool sw = tau > 0.0f;                                      // de Rijk: larger diagonal to p
    // X[:,p] <- C x_p - S x_q ; X[:,q] <- S x_p + C x_q   (same convention as the tile kernels)
    const float C2 = sw ? s0 : c0, S2 = sw ? -c0 : -s0;
    // ---- G <- J^T G J on the thread's 2x2 blocks (rows of pair kr + 8 i, columns of pair k2) ----
#pragma unroll
    for (int i = 0; i < INNER_NB; ++i) {
      const int la = 2 * wv_s + INNER_KR * i;                        // wave-uniform lane holding the row pair's rotation
      const float C1a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C2), la));
      const float C1b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C2), la + 1));
      const float S1a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S2), la));
      const float S1b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S2), la + 1));
      const float C1 = hi ? C1b : C1a, S1 = hi ? S1b : S1a;
      const float gpp = g[i][0], gpq = g[i][1], gqp = g[i][2], gqq = g[i][3];
      const float a0 = C2 * gpp - S2 * gpq, a1 = S2 * gpp + C2 * gpq;   // row p1, columns rotated
      const float b0 = C2 * gqp - S2 * gqq, b1 = S2 * gqp + C2 * gqq;   // row q1
      Gn[p1[i]][p2] = C1 * a0 - S1 * b0; Gn[p1[i]][q2] = C1 * a1 - S1 * b1;    // rows rotated
      Gn[q1[i]][p2] = S1 * a0 + C1 * b0; Gn[q1[i]][q2] = S1 * a1 + C1 * b1;
    }
    // ---- R <- R J : rows kr + 8 j, column pair k2 ----
#pragma unroll
    for (int j = 0; j < 2 * INNER_NB; ++j) {
      const int r = kr + INNER_KR * j;
      R[r][p2] = C2 * rp_[j] - S2 * rq_[j]; R[r][q2] = S2 * rp_[j] + C2 * rq_[j];
    }
    __syncthreads();
    G = Gn;
  }
#if defined(WM_INNER_DIAG)
  st3 = __builtin_amdgcn_s_memtime();
#endif
  for (int e = t; e < RP * RP; e += INNER_NT) {
    out[h_units ? pk_index(e >> 6, e & 63) : e] = R[e >> 6][e & 63];
    if (rpk) put_r(e >> 6, e & 63, R[e >> 6][e & 63]);
  }
  if (h_units) writeback(&G[0][0]);
#if defined(WM_INNER_DIAG)
  __syncthreads();
  if (t == 0 && p == 1 && blockIdx.z == 0) {
    const unsigned long long st4 = __builtin_amdgcn_s_memtime();
    printf("inner diag: load %llu maxcos %llu loop %llu (%d steps, %llu per step) store %llu cycles\n", st1 - st0, st2 - st1,
           st3 - st2, n_inner, (st3 - st2) / (unsigned long long)n_inner, st4 - st3);
  }
#endif
}

// ---------------------------------------------------------------------------
// Aug rows of the pair <- R^T x rows, one 64-column tile per workgroup.
// ---------------------------------------------------------------------------
// MFMA form (v_mfma_f32_32x32x2_f32; operand layout checked by tools/mfma_probe.hip):
//   out[i][c] = sum_k R[k][i] X[k][c],  i, k in 0..63 (the pair's rows), c = a 32-column block.
// A wave owns 32 columns: it loads the 64 x 32 block of X straight from global memory into the B
// operands (lane j + 32 h holds X[2m + h][c0 + j] for m = 0..31: two coalesced 128-byte rows per
// load), takes the A operands R[2m + h][i0 + j] from the LDS copy of R, and accumulates both row
// halves (i0 = 0, 32) in 2 x 16 accumulator registers: 64 MFMAs per block, no LDS traffic for X,
// in place (the block is fully in registers before the first store).
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_rf_apply(float* __restrict__ aug, const size_t aug_plane_stride,
                                                 const int ld, const int ncols, const int blocks_per_wave,
                                                 const int2* __restrict__ pairs, const float* __restrict__ Rall,
                                                 const int* __restrict__ skip_flags) {
  __shared__ float Rs[RP][RP + 1];      // [k][i]
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, j = lane & 31, h = lane >> 5;
  const int p = blockIdx.x;
  if (skip_flags[(size_t)blockIdx.z * gridDim.x + p]) return;      // the inner solve found nothing to rotate
  aug += (size_t)blockIdx.z * aug_plane_stride;
  const float* Rp = Rall + ((size_t)blockIdx.z * gridDim.x + p) * RP * RP;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = t + 256 * i;
    Rs[e >> 6][e & 63] = Rp[e];
  }
  __syncthreads();
  const int2 pr = pairs[p];
  const size_t row_x = (size_t)(pr.x * RB + h) * ld, row_y = (size_t)(pr.y * RB + h) * ld;
  const int blk0 = (blockIdx.y * 4 + wv) * blocks_per_wave;
  for (int bk = 0; bk < blocks_per_wave; ++bk) {
    const int c0 = (blk0 + bk) * 32;
    if (c0 >= ncols) break;                        // wave-uniform
    const int col = c0 + j;
    const bool valid = col < ncols;
    const int colc = valid ? col : ncols - 1;      // loads stay unconditional (a clamped column): per-element
    const float* px = aug + row_x + colc;          // selects turn into branches that serialise the 32 loads
    const float* py = aug + row_y + colc;
    float x[32];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      x[m] = px[(size_t)(2 * m) * ld];
      x[16 + m] = py[(size_t)(2 * m) * ld];
    }
    v16f acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const float a0 = Rs[2 * m + h][j], a1 = Rs[2 * m + h][32 + j];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x[m], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x[m], acc1, 0, 0, 0);
    }
    if (valid) {
      float* ox = aug + (size_t)(pr.x * RB) * ld + col;
      float* oy = aug + (size_t)(pr.y * RB) * ld + col;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * h + (v % 4);     // accumulator register v, lane half h -> output row
        ox[(size_t)i * ld] = acc0[v];
        oy[(size_t)i * ld] = acc1[v];
      }
    }
  }
}

// squared norms of the B part (first M columns) and the Qt part of every row
__global__ __launch_bounds__(256) void k_rf_rownorms(const float* __restrict__ aug, const size_t aug_plane_stride,
                                                    const int ld, const int M, const int Lq,
                                                    double* __restrict__ b2, double* __restrict__ q2) {
  __shared__ double red[2][4];
  const int i = blockIdx.x, t = threadIdx.x;
  const int Lp = Lq;
  aug += (size_t)blockIdx.y * aug_plane_stride;
  b2 += (size_t)blockIdx.y * gridDim.x; q2 += (size_t)blockIdx.y * gridDim.x;
  const float* row = aug + (size_t)i * ld;
  double sb = 0.0, sq = 0.0;
  for (int j = t; j < M; j += 256) sb += (double)row[j] * (double)row[j];
  for (int j = t; j < Lp; j += 256) sq += (double)row[M + j] * (double)row[M + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sb += __shfl_down(sb, o, 64); sq += __shfl_down(sq, o, 64); }
  if ((t & 63) == 0) { red[0][t >> 6] = sb; red[1][t >> 6] = sq; }
  __syncthreads();
  if (t == 0) {
    b2[i] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q2[i] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// t2[i] = sum_j T[j][i]^2 (float64) of the dense [rows x cols] matrix T of every plane
__global__ __launch_bounds__(256) void k_rf_colnorms(const float* __restrict__ T, const size_t plane_stride,
                                                    const int rows, const int cols, double* __restrict__ t2) {
  __shared__ double red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6;
  T += (size_t)blockIdx.y * plane_stride;
  double acc = 0.0;
  if (c < cols)
    for (int j = rs; j < rows; j += 4) { const double v = (double)T[(size_t)j * cols + c]; acc += v * v; }
  red[rs][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rs == 0 && c < cols) t2[(size_t)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// rows of the B part scaled in place:  B[i][:] *= d[i]
__global__ void k_rf_scale_rows(float* __restrict__ aug, const size_t aug_plane_stride, const int ld,
                                const int M, const float* __restrict__ d) {
  const int i = blockIdx.y;
  aug += (size_t)blockIdx.z * aug_plane_stride;
  const float s = d[(size_t)blockIdx.z * gridDim.y + i];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x)
    aug[(size_t)i * ld + j] *= s;
}

// Yw (float, logical A layout L x M or its transpose) -> clip/truncate -> uint8 plane; optional float copy
__global__ void k_rf_quant(const float* __restrict__ yw, const size_t yw_plane_stride, const int ldy,
                           const int transpose, uint8_t* __restrict__ dst, const size_t dst_stride,
                           const size_t dst_plane_stride, float* __restrict__ ywout, const int H, const int W) {
  const int r = blockIdx.y;
  yw += (size_t)blockIdx.z * yw_plane_stride;
  dst += (size_t)blockIdx.z * dst_plane_stride;
  if (ywout) ywout += (size_t)blockIdx.z * (size_t)H * W;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W; c += gridDim.x * blockDim.x) {
    const float v = transpose ? yw[(size_t)c * ldy + r] : yw[(size_t)r * ldy + c];
    if (ywout) ywout[(size_t)r * W + c] = v;
    dst[(size_t)r * dst_stride + c] = (uint8_t)(unsigned)fminf(fmaxf(v, 0.0f), 255.0f);
  }
}

// gather sorted, normalised factors:  out[k][:] = src_row[order[k]][:] * scale[k]
__global__ void k_rf_gather_rows(const float* __restrict__ src, const int ld, const int ncols,
                                 const int* __restrict__ order, const float* __restrict__ scale,
                                 float* __restrict__ dst, const int ldd) {
  const int k = blockIdx.y;
  const float* row = src + (size_t)order[k] * ld;
  const float s = scale[k];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += gridDim.x * blockDim.x)
    dst[(size_t)k * ldd + j] = row[j] * s;
}

// the sorted, normalised factors of all planes of a batch (grid.z = plane):
//   straight:    dst[z][k][c] = src[z][order[z][k]][c] * scale[z][k]          (grid: column blocks x rows k)
//   transposing: dst[z][c][k] = src[z][order[z][k]][c] * scale[z][k]          (32 x 32 tiles through LDS, both sides coalesced)
__global__ void k_rf_gather_rows_b(const float* __restrict__ src, const size_t src_ps, const int ld, const int ncols,
                                   const int* __restrict__ order, const float* __restrict__ scale, const int os,
                                   float* __restrict__ dst, const size_t dst_ps, const int ldd) {
  const int k = blockIdx.y, z = blockIdx.z;
  const float* row = src + (size_t)z * src_ps + (size_t)order[(size_t)z * os + k] * ld;
  const float sc = scale[(size_t)z * os + k];
  float* d = dst + (size_t)z * dst_ps + (size_t)k * ldd;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += gridDim.x * blockDim.x) d[j] = row[j] * sc;
}

__global__ __launch_bounds__(256) void k_rf_gather_rows_t(const float* __restrict__ src, const size_t src_ps, const int ld, const int ncols,
                                                         const int nk, const int* __restrict__ order, const float* __restrict__ scale,
                                                         const int os, float* __restrict__ dst, const size_t dst_ps, const int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, z = blockIdx.z;
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  src += (size_t)z * src_ps; dst += (size_t)z * dst_ps; order += (size_t)z * os; scale += (size_t)z * os;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (k < nk && c < ncols) ? src[(size_t)order[k] * ld + c] * scale[k] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, k = k0 + tx;
    if (c < ncols && k < nk) dst[(size_t)c * ldd + k] = tile[tx][ty + 8 * i];
  }
}

// deterministic pseudo-random pattern in (-1, 1): the start vectors of the null-space completion
// (rank-deficient planes); element (r, c) of the matrix with seed `seed`
__global__ void k_rf_pattern(float* __restrict__ dst, const int cols, const unsigned seed) {
  const int r = blockIdx.y;          // grid.y = rows
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += gridDim.x * blockDim.x) {
    unsigned h = (unsigned)r * 0x9E3779B1u ^ ((unsigned)c + seed) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    dst[(size_t)r * cols + c] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
  }
}

// dst[r][k] = src[r][k] * d[k]   (rows x cols, dense, batch = grid.z with strides)
__global__ void k_rf_scale_cols_b(const float* __restrict__ src, const size_t s_src, float* __restrict__ dst,
                                  const size_t s_dst, const int cols, const float* __restrict__ d, const size_t s_d) {
  const int r = blockIdx.y;
  src += (size_t)blockIdx.z * s_src; dst += (size_t)blockIdx.z * s_dst; d += (size_t)blockIdx.z * s_d;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cols; k += gridDim.x * blockDim.x)
    dst[(size_t)r * cols + k] = src[(size_t)r * cols + k] * d[k];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct RefPlan {
  int H, W, L, M, Lp, ld, nbk, npairs, nsteps, nch, B;
  int apply_tiles;       // 32-column blocks per wave of an apply workgroup (4 waves)
  size_t aug_ps;         // floats between the Aug matrices of consecutive planes
  bool transpose;        // A = plane^T (portrait planes: the short side must index rows)
};

RefPlan make_plan(int H, int W, int B = 1) {
  RefPlan p;
  p.H = H; p.W = W; p.B = B;
  p.transpose = H > W;
  p.L = std::min(H, W); p.M = std::max(H, W);
  p.Lp = (p.L + RP - 1) / RP * RP;
  p.ld = (p.M + p.Lp + 3) & ~3;
  p.aug_ps = (size_t)p.Lp * p.ld;
  p.nbk = p.Lp / RB; p.npairs = p.nbk / 2; p.nsteps = p.nbk - 1;
  // Work per workgroup grows with the batch: with many planes in a launch the grid is large anyway,
  // so a workgroup takes more columns (R / partial-sum traffic amortised, loads pipelined); a single
  // plane keeps the small units that fill the chip.
  // (round 2 gave a workgroup two 32-column blocks per wave from 96 (pair, plane) items per step on; with the batch split
  // into two or three plane groups a launch rarely gets there, and one block per wave measures the same or better at every
  // batch size: 8 planes 106.8 / 106.4, 16 planes 139.3 / 137.2, 24 planes 140.4 / 139.4 frames/s - profiles/r03z_fullframe_queues.log)
  p.apply_tiles = 1;
  if (const char* e = getenv("WM_RF_APPLY_TILES")) { const int v = atoi(e); if (v >= 1 && v <= 8) p.apply_tiles = v; }   // tuning knob
  p.nch = (p.M + GRAM_CC - 1) / GRAM_CC;    // Gram stays in many small units: a float4 / double-buffered / 512-column variant measured slower (36-40 vs 31 us)
  return p;
}

struct RefWs {           // carved out of a grow-only context buffer; every per-plane array is [B][...]
  float* aug; float* partials; float* R; const int2* pairs; unsigned* maxcos; float* floor2; int* skip; double* b2; double* q2;
  float* dvec; int* order; float* scale; float* tmp1; float* tmp2;
};

inline size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

// The round-robin tournament of a plan depends on its block count only: built once per context and
// block count (a handful of sizes per process), kept on the device, handed out without a copy or a sync.
int get_pairs(wm_ctx* ctx, const RefPlan& p, const int2** out) {
  for (int i = 0; i < wm_ctx::MAX_PAIR_TABS; ++i)
    if (ctx->pair_tab[i] && ctx->pair_tab_nbk[i] == p.nbk) { *out = (const int2*)ctx->pair_tab[i]; return WM_OK; }
  int slot = -1;
  for (int i = 0; i < wm_ctx::MAX_PAIR_TABS; ++i) if (!ctx->pair_tab[i]) { slot = i; break; }
  if (slot < 0) {                                   // full: recycle round-robin (nothing in flight may still read it)
    slot = ctx->pair_tab_next; ctx->pair_tab_next = (ctx->pair_tab_next + 1) % wm_ctx::MAX_PAIR_TABS;
    WM_HIP(hipStreamSynchronize(ctx->stream));
    WM_HIP(hipFree(ctx->pair_tab[slot]));
    ctx->pair_tab[slot] = nullptr;
  }
  std::vector<int2> tab((size_t)p.nsteps * p.npairs);
  std::vector<int> idx(p.nbk);
  std::iota(idx.begin(), idx.end(), 0);
  for (int s = 0; s < p.nsteps; ++s) {
    for (int k = 0; k < p.npairs; ++k) {
      const int a = idx[k], b = idx[p.nbk - 1 - k];
      tab[(size_t)s * p.npairs + k] = make_int2(std::min(a, b), std::max(a, b));
    }
    const int last = idx[p.nbk - 1];
    for (int j = p.nbk - 1; j > 1; --j) idx[j] = idx[j - 1];
    idx[1] = last;
  }
  if (hipMalloc(&ctx->pair_tab[slot], std::max<size_t>(tab.size(), 1) * sizeof(int2)) != hipSuccess) {
    (void)hipGetLastError();
    ctx->pair_tab[slot] = nullptr;
    return set_err(WM_ERR_NOMEM, "hipMalloc failed for %s", "pair table");
  }
  if (!tab.empty()) {
    WM_HIP(hipMemcpyAsync(ctx->pair_tab[slot], tab.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));     // tab is a local
  }
  ctx->pair_tab_nbk[slot] = p.nbk;
  *out = (const int2*)ctx->pair_tab[slot];
  return WM_OK;
}

// which = 0: the context's main full-frame workspace; 1: the second one (null-space completion, so that the
// first call's arrays stay where they are)
int plan_workspace(wm_ctx* ctx, const RefPlan& p, RefWs& w, size_t extra_f32_a, size_t extra_f32_b, int which = 0) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += a256(bytes); return o; };
  const size_t B = (size_t)p.B;
  const size_t o_aug = take(B * p.aug_ps * 4), o_par = take(B * p.npairs * p.nch * GRAM_PART * 4),
               o_R = take(B * p.npairs * RP * RP * 4),
               o_mc = take(B * 4 + 256), o_fl = take(B * 4 + 256), o_skip = take(B * p.npairs * 4 + (size_t)(2 + wm_ctx::MAX_AUX) * 256), o_b2 = take(B * p.Lp * 8), o_q2 = take(B * p.Lp * 8),
               o_d = take(2 * B * p.Lp * 4), o_ord = take((size_t)p.Lp * 4), o_sc = take((size_t)p.Lp * 4),
               o_t1 = take(extra_f32_a * 4), o_t2 = take(extra_f32_b * 4);
  if (which == 0) WM_TRY(grow(ctx, &ctx->ref_ws, &ctx->ref_ws_bytes, off, "full-frame workspace"));
  else WM_TRY(grow(ctx, &ctx->ref_ws2, &ctx->ref_ws2_bytes, off, "full-frame completion workspace"));
  char* b = (char*)(which == 0 ? ctx->ref_ws : ctx->ref_ws2);
  w.aug = (float*)(b + o_aug); w.partials = (float*)(b + o_par); w.R = (float*)(b + o_R);
  w.maxcos = (unsigned*)(b + o_mc); w.floor2 = (float*)(b + o_fl); w.skip = (int*)(b + o_skip); w.b2 = (double*)(b + o_b2);
  w.q2 = (double*)(b + o_q2); w.dvec = (float*)(b + o_d); w.order = (int*)(b + o_ord);
  w.scale = (float*)(b + o_sc); w.tmp1 = (float*)(b + o_t1); w.tmp2 = (float*)(b + o_t2);
  WM_TRY(get_pairs(ctx, p, &w.pairs));
  return WM_OK;
}

// orthonormal DCT-II basis D_n (float64 on the host, rounded to float32) into cache slot `slot`
int fill_dct(wm_ctx* ctx, int n, int slot) {
  if (ctx->dct_mat[slot]) {
    WM_HIP(hipStreamSynchronize(ctx->stream));
    WM_HIP(hipFree(ctx->dct_mat[slot]));
    ctx->dct_mat[slot] = nullptr; ctx->dct_n[slot] = 0;
  }
  if (hipMalloc((void**)&ctx->dct_mat[slot], (size_t)2 * n * n * 4) != hipSuccess) {       // D, then D^T (row-major A operand of the split-f16 inverse DCT)
    (void)hipGetLastError();
    return set_err(WM_ERR_NOMEM, "hipMalloc failed for %s", "DCT basis");
  }
  std::vector<float> D((size_t)2 * n * n);
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < n; ++k) {
    const double s = (k == 0) ? sqrt(1.0 / n) : sqrt(2.0 / n);
    for (int m = 0; m < n; ++m) D[(size_t)n * n + (size_t)m * n + k] = D[(size_t)k * n + m] = (float)(s * cos(pi * (2.0 * m + 1.0) * k / (2.0 * n)));
  }
  WM_HIP(hipMemcpyAsync(ctx->dct_mat[slot], D.data(), D.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  ctx->dct_n[slot] = n;
  return WM_OK;
}

// D_H and D_W of one plane, from the two-slot per-context cache.  Both are resolved in ONE
// call: a slot that already holds one of the two sizes is never the one evicted for the other
// (fetching them one after the other could free the matrix just handed out - found by the
// re-entrancy test, which ran 96x128 on a context whose cache held 64 and 96).
int get_dct_pair(wm_ctx* ctx, int H, int W, float** dH, float** dW) {
  int sh = -1, sw = -1;
  for (int i = 0; i < 2; ++i) {
    if (ctx->dct_mat[i] && ctx->dct_n[i] == H && sh < 0) sh = i;
    if (ctx->dct_mat[i] && ctx->dct_n[i] == W && sw < 0) sw = i;
  }
  if (sh < 0) {
    sh = (sw == 0) ? 1 : 0;                      // any slot that does not hold W
    WM_TRY(fill_dct(ctx, H, sh));
    if (H == W) sw = sh;
  }
  if (sw < 0) {
    sw = 1 - sh;                                 // the slot that does not hold H
    WM_TRY(fill_dct(ctx, W, sw));
  }
  *dH = ctx->dct_mat[sh];
  *dW = ctx->dct_mat[sw];
  return WM_OK;
}

#include "wm_ref_hier.inc"

// block one-sided Jacobi on the B Aug matrices (already loaded); every launch covers
// a group of planes (grid.z), sweeps continue until every plane's Gram matrices are
// diagonal to CONV_COS.  sweeps_out: sweeps used (negative: bound hit).
//
// A step is gram -> inner -> apply, and k_rf_inner is one latency-bound workgroup per
// block pair (~35 us with most of the chip idle).  With two or more planes the batch is
// split into two groups on two HIP streams, the second started one gram later, so one
// group's inner solve runs under the other group's gram/apply tiles.
// what the rotated rows are needed for: their norms only (extract / detect), their directions too (embed:
// u_i, v_i enter the stego), or the accumulated left factor as well (watermark-side SVD, [A | I])
// JR_ORTH: orthogonalise the rows only (null-space completion): no left factor, the tight stopping cosine
enum JacobiUse { JR_SIGMA, JR_EMBED, JR_SVD, JR_ORTH };

int jacobi_rows(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const JacobiUse use, int* sweeps_out) {
  const bool with_q = use == JR_SVD, sigma_only = use == JR_SIGMA;
  const int ncols = with_q ? p.M + p.Lp : p.M;
  int sweep = 0;
  bool done = false;
  std::vector<unsigned> bits(p.B);
  static const int full_sweeps = getenv("WM_RF_FULL_SWEEPS") ? atoi(getenv("WM_RF_FULL_SWEEPS")) : FULL_INNER_SWEEPS;
  // number of plane groups (HIP queues): WM_RF_QUEUES=n (1..4), WM_RF_ONE_QUEUE=1 is the old spelling of n = 1
  // default: two groups, three from 12 planes on (16 planes: 131 -> 137 frames/s on two boxes, 24: 138 -> 139; 8 planes: 107 / 106)
  const int env_queues = [] {          // (read on every call)
    if (getenv("WM_RF_ONE_QUEUE") && atoi(getenv("WM_RF_ONE_QUEUE"))) return 1;
    const int n = getenv("WM_RF_QUEUES") ? atoi(getenv("WM_RF_QUEUES")) : 0;
    return n < 1 ? 0 : (n > 1 + wm_ctx::MAX_AUX ? 1 + wm_ctx::MAX_AUX : n);
  }();
  const int max_queues = env_queues ? env_queues : std::min(1 + wm_ctx::MAX_AUX, p.B >= 12 ? DEFAULT_QUEUES + 1 : DEFAULT_QUEUES);
  static const float conv_sigma = getenv("WM_RF_CONV_SIGMA") ? (float)atof(getenv("WM_RF_CONV_SIGMA")) : CONV_COS_SIGMA;
  // The accumulated factor of JR_SVD needs every rotation; the other two uses stop earlier: a residual
  // cosine c moves the stego by c * s_max / s_i of a (sub-LSB) term and the singular values by the
  // bounds documented at fetch_norms_t, the same for embed and extract so that it cancels in S_cw - Sc.
  const float conv_cos = (with_q || use == JR_ORTH) ? CONV_COS : conv_sigma;
  const float skip_thr = SKIP_FRACTION * conv_cos;
  (void)sigma_only;
  ctx->ref_skip_thr = skip_thr;
  const int NQ = std::min(max_queues, p.B);               // group g owns planes [zb[g], zb[g + 1])
  int zb[2 + wm_ctx::MAX_AUX];
  for (int g = 0; g <= NQ; ++g) zb[g] = (int)((long long)g * p.B / NQ);
  for (int g = 1; g < NQ; ++g)
    if (!ctx->aux_stream[g - 1]) {
      WM_HIP(hipStreamCreateWithFlags(&ctx->aux_stream[g - 1], hipStreamNonBlocking));
      WM_HIP(hipEventCreateWithFlags(&ctx->ev_fork[g - 1], hipEventDisableTiming));
      WM_HIP(hipEventCreateWithFlags(&ctx->ev_join[g - 1], hipEventDisableTiming));
    }
  // numerical-null floor per plane: (NULL_ROW_RATIO * |A|_F)^2 from the row norms of the loaded input
  {
    hipLaunchKernelGGL(k_rf_rownorms, dim3(p.Lp, p.B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld, p.M, 0, w.b2, w.q2);
    std::vector<double> r2((size_t)p.B * p.Lp);
    WM_HIP(hipMemcpyAsync(r2.data(), w.b2, r2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<float> fl(p.B);
    for (int z = 0; z < p.B; ++z) {
      double f2 = 0.0;
      for (int i = 0; i < p.Lp; ++i) f2 += r2[(size_t)z * p.Lp + i];
      fl[z] = (float)(NULL_ROW_RATIO * NULL_ROW_RATIO * f2);
    }
    WM_HIP(hipMemcpyAsync(w.floor2, fl.data(), fl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));       // fl is a local
  }
  const size_t par_ps = (size_t)p.npairs * p.nch * GRAM_PART, r_ps = (size_t)p.npairs * RP * RP;
  auto step = [&](hipStream_t st, int g, int s, int part) {
    const int z0 = zb[g], nz = zb[g + 1] - zb[g];
    const int2* pr = w.pairs + (size_t)s * p.npairs;
    float* aug = w.aug + (size_t)z0 * p.aug_ps;
    float* par = w.partials + (size_t)z0 * par_ps;
    float* R = w.R + (size_t)z0 * r_ps;
    if (part & 1)
      hipLaunchKernelGGL(k_rf_gram, dim3(p.npairs, p.nch, nz), dim3(256), 0, st, aug, p.aug_ps, p.ld, p.M, pr, par);
    if (part & 2) {
      hipLaunchKernelGGL(k_rf_inner, dim3(p.npairs, 1, nz), dim3(INNER_NT), 0, st, par, p.nch, R, w.maxcos + z0,
                         w.floor2 + z0, (s == 0 || sweep < full_sweeps) ? 0 : 1, hier_flag_base(w.skip, (size_t)p.npairs, z0, g),
                         skip_thr, (const int*)nullptr, (float*)nullptr, (int*)nullptr, 0, (float*)nullptr, (int*)nullptr, 0);
      const int n_blk = (ncols + 31) / 32, per_wg = 4 * p.apply_tiles;     // 32-column blocks, 4 waves per workgroup
      hipLaunchKernelGGL(k_rf_apply, dim3(p.npairs, (n_blk + per_wg - 1) / per_wg, nz), dim3(256), 0, st,
                         aug, p.aug_ps, p.ld, ncols, p.apply_tiles, pr, R, hier_flag_base(w.skip, (size_t)p.npairs, z0, g));
    }
  };
  // Two-level scheme (wm_ref_hier.inc): the same rotations with the rows streamed 3 times per SUPER-step.
  // WM_RF_HIER=0 / 1 forces the flat tournament / the two-level scheme (read per call, so that a test can hold the two against each other);
  // WM_RF_HIER_SB = blocks per super-block (2, 4 or 6).
  // Default (measured on 1080p planes, profiles/r04_hier_crossover.log): with the split-f16 kernels - uint8 planes, i.e. every
  // embed / sigma / extract / detect call - the two-level scheme wins from 4 planes per call on (2: 44 / 44, 4: 72 / 79, 8: 104 / 128,
  // 16: 138 / 180, 64: 140 / 242 frames/s flat / two-level) and is used from HIER_MIN_PLANES_F16 = 3; with the f32 kernels (float
  // inputs: the watermark-side SVD, the completion) only from HIER_MIN_PLANES = 20 (2 planes: 20.9 / 26.5 ms - a small batch is
  // latency-bound and the flat step's chain gram - inner - apply is the shorter one).
  const int f16_env = getenv("WM_RF_HIER_F16") ? atoi(getenv("WM_RF_HIER_F16")) : HIER_F16_DEFAULT;     // bit 0: Gram, bit 1: rotation products
  const bool f16_ok = use == JR_SIGMA || use == JR_EMBED;     // rows of uint8 planes: |entries| <= 255 sqrt(L) < 65 504 under orthogonal rotations
  const bool gram_f16 = (f16_env & 1) && f16_ok, apply_f16 = (f16_env & 2) && f16_ok;
  const bool hier = getenv("WM_RF_HIER") ? atoi(getenv("WM_RF_HIER")) != 0
                                         : p.B >= ((gram_f16 && apply_f16) ? HIER_MIN_PLANES_F16 : HIER_MIN_PLANES);
  const HierTab* ht = nullptr;
  HierWs hw{};
  const int hdbg = getenv("WM_RF_HDBG") ? atoi(getenv("WM_RF_HDBG")) : 0;
  // Gram tiles on the f16 matrix pipe with split operands (k_hgram_h; WM_RF_HIER_F16=0: the f32 form).  Row entries are bounded
  // by 255 sqrt(L): beyond L = 65 536 (never a plane) f16 would overflow, and float inputs (the watermark-side SVD of a DCT
  // plane, the completion's random vectors) have no such bound - those uses keep the f32 kernel.
  if (hier) {
    int sb = 6;
    if (const char* e = getenv("WM_RF_HIER_SB")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 6) sb = v; }
    WM_TRY(get_hier(ctx, p.nbk, sb, &ht));
    WM_TRY(plan_hier_ws(ctx, p, *ht, (p.B + NQ - 1) / NQ, hw));
    static bool attr_set = false;
    if (!attr_set) {
      WM_HIP(hipFuncSetAttribute((const void*)k_happly, hipFuncAttributeMaxDynamicSharedMemorySize, HN * 65 * 4));
      WM_HIP(hipFuncSetAttribute((const void*)k_happly_h<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * HA_CP * 2));
      attr_set = true;
    }
  }
  auto hstep = [&](hipStream_t st, int g, int s1) {
    const int z0 = zb[g], nz = zb[g + 1] - zb[g], nsp = ht->nsp;
    const HSuper* sup = ht->d_super + (size_t)s1 * nsp;
    float* aug = w.aug + (size_t)z0 * p.aug_ps;
    float* Gs = hw.Gs + (size_t)z0 * nsp * HN * HN;
    float* R = hw.R + (size_t)z0 * nsp * HU * RP * RP;
    float* par = hw.partials + (size_t)z0 * nsp * HG_TILES * hw.KS * HG_T * HG_T;
    int* skip = hier_flag_base(hw.skip, (size_t)nsp * HU, z0, g);
    int* anyrot = hier_flag_base(hw.anyrot, (size_t)nsp, z0, g);
    // per-stage rotations (packed) and skip flags of this group's planes: [stage][plane][super-pair * HU + unit]
    float* Rpk = hw.Rpk + (size_t)z0 * HT_MAX * nsp * HU * RP * RP;
    int* skipT = hier_flag_base(hw.skipT, (size_t)HT_MAX * nsp * HU, z0, g);
    const size_t rpk_stage = (size_t)nz * nsp * HU * RP * RP, skip_stage = (size_t)nz * nsp * HU;
    const int n32 = ht->nmax * RB, npmax = (n32 + HG_T - 1) / HG_T;
    const int nchunk = (p.M + HG_KC - 1) / HG_KC;
    // (one workgroup per (super-pair, plane, split) instead of three: only where that still fills the chip - from 11 planes per launch on;
    //  16 planes on three queues: 209 against 213 frames/s with it, 64 planes: 308 against 297)
    const bool h3 = getenv("WM_RF_HGRAM3") ? atoi(getenv("WM_RF_HGRAM3")) != 0 : nsp * hw.KS * nz >= 256;
    const bool use_h3 = gram_f16 && npmax == 3 && h3;     // (its column splits weigh more: every split writes the whole upper triangle - 2.0 against 0.25: +2 % at 64 and 128 planes)
    const int KS = hier_ks(nsp, nz, use_h3 ? 1 : npmax, nchunk, hw.KS, use_h3 ? 2.0 : 0.25), cps = (nchunk + KS - 1) / KS;
    const int ngrp = nsp * KS * nz;               // (super-pair, split, plane) groups of npmax workgroups, dealt over the XCDs
    if (gram_f16 && npmax == 3 && h3)               // three panels: the rows fetched once per chunk (k_hgram_h3)
      hipLaunchKernelGGL(k_hgram_h3, dim3(((ngrp + 7) / 8) * 8), dim3(H3_NT), 0, st, aug, p.aug_ps, p.ld, p.M, sup, nsp,
                         par, KS, cps, nz);
    else if (gram_f16)
      hipLaunchKernelGGL(k_hgram_h, dim3(((ngrp + 7) / 8) * 8 * npmax), dim3(512), 0, st, aug, p.aug_ps, p.ld, p.M, sup, nsp,
                         par, KS, cps, npmax, nz);
    else
      hipLaunchKernelGGL(k_hgram, dim3(((ngrp + 7) / 8) * 8 * npmax), dim3(512), 0, st, aug, p.aug_ps, p.ld, p.M, sup, nsp,
                         par, KS, cps, npmax, nz, hdbg);
    hipLaunchKernelGGL(k_hreduce, dim3(HSB * (HSB + 1) / 2, nsp, nz), dim3(256), 0, st, par, sup, nsp, KS, Gs, anyrot);
    constexpr int NG = HU * (HU - 1) / 2;
    const int T = ht->T[s1];
    const int* un0 = ht->d_units + (size_t)ht->stage_off[s1] * nsp * HU;
    for (int t = 0; t < T; ++t) {
      const int* un = un0 + (size_t)t * nsp * HU;
      hipLaunchKernelGGL(k_rf_inner, dim3(nsp * HU, 1, nz), dim3(INNER_NT), 0, st, (const float*)nullptr, 0, R, w.maxcos + z0,
                         w.floor2 + z0, (s1 == 0 && t == 0) ? 0 : 1, skip, skip_thr, un, Gs, anyrot, nsp, Rpk + t * rpk_stage,
                         skipT + t * skip_stage, apply_f16 ? 1 : 0);
      if (t + 1 < T)                               // nothing reads G_s after the last stage
        hipLaunchKernelGGL(k_hupdate, dim3(NG, nsp, nz), dim3(256), 0, st, un, sup, nsp, R, skip, Gs);
    }
    const int ntask = nsp * nz * ((ncols + 63) / 64);
    if (apply_f16)
      hipLaunchKernelGGL((k_happly_h<64>), dim3(8 * ((ntask + 7) / 8)), dim3(64 * ht->nmax), (size_t)2 * 64 * HA_CP * 2, st, aug, p.aug_ps, p.ld,
                         ncols, sup, nsp, nz, un0, T, Rpk, skipT, rpk_stage, skip_stage, anyrot, hdbg >> 4);
    else
      hipLaunchKernelGGL(k_happly, dim3(8 * ((ntask + 7) / 8)), dim3(64 * ht->nmax), (size_t)n32 * 65 * 4, st, aug, p.aug_ps, p.ld, ncols,
                         sup, nsp, nz, un0, T, Rpk, skipT, rpk_stage, skip_stage, anyrot, hdbg >> 4);
  };
  while (!done && sweep < MAX_SWEEPS) {
    WM_HIP(hipMemsetAsync(w.maxcos, 0, (size_t)p.B * sizeof(unsigned), ctx->stream));
    // group g starts one gram after group g - 1, so that the groups' latency-bound inner solves
    // fall under each other's gram / apply tiles instead of all at the same time
    auto q = [&](int g) { return g == 0 ? ctx->stream : ctx->aux_stream[g - 1]; };
    if (hier) {
      for (int g = 1; g < NQ; ++g) {               // the other queues start behind the memset
        if (g == 1) WM_HIP(hipEventRecord(ctx->ev_fork[0], ctx->stream));
        WM_HIP(hipStreamWaitEvent(q(g), ctx->ev_fork[0], 0));
      }
      for (int s1 = 0; s1 < ht->nsteps1; ++s1)
        for (int g = 0; g < NQ; ++g) hstep(q(g), g, s1);
    } else {
    for (int g = 0; g < NQ; ++g) {
      if (g > 0) WM_HIP(hipStreamWaitEvent(q(g), ctx->ev_fork[g - 1], 0));
      step(q(g), g, 0, 1);
      if (g + 1 < NQ) WM_HIP(hipEventRecord(ctx->ev_fork[g], q(g)));
    }
    for (int g = 0; g < NQ; ++g) step(q(g), g, 0, 2);
    for (int s = 1; s < p.nsteps; ++s)
      for (int g = 0; g < NQ; ++g) step(q(g), g, s, 3);
    }
    for (int g = 1; g < NQ; ++g) {
      WM_HIP(hipEventRecord(ctx->ev_join[g - 1], q(g)));
      WM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[g - 1], 0));
    }
    WM_HIP(hipGetLastError());
    WM_HIP(hipMemcpyAsync(bits.data(), w.maxcos, (size_t)p.B * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
    ++sweep;
    done = true;
    for (int z = 0; z < p.B; ++z) { float mc; memcpy(&mc, &bits[z], 4); if (!(mc < conv_cos)) done = false; }
  }
  *sweeps_out = done ? sweep : -sweep;
  ctx->ref_last_sweeps = sweep;
  {   // nominal matrix-core flops of one sweep (every pair counted as rotated), for bench.py's roofline
    double f = 0.0;
    if (hier) {
      for (int s1 = 0; s1 < ht->nsteps1; ++s1) {
        // the host copy of the tables holds the stage counts; the super-pairs' sizes follow from the block counts
        f += ht->flops_step[s1] * ((double)p.M) + ht->flops_apply_step[s1] * (double)ncols + ht->flops_stage_step[s1];
      }
    } else {
      f = (double)p.nsteps * p.npairs * (2.0 * 3 * RB * RB * p.M + 2.0 * RP * RP * ncols);
    }
    ctx->ref_last_flops = f * sweep * p.B;
    ctx->ref_last_hier = hier ? 1 : 0;
  }
  return WM_OK;
}

// row norms of every plane -> host (b2, q2 are [B][Lp])
int fetch_norms(wm_ctx* ctx, const RefPlan& p, const RefWs& w, bool with_q, std::vector<double>& b2,
                std::vector<double>& q2) {
  hipLaunchKernelGGL(k_rf_rownorms, dim3(p.Lp, p.B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld, p.M,
                     with_q ? p.Lp : 0, w.b2, w.q2);
  WM_HIP(hipGetLastError());
  b2.resize((size_t)p.B * p.Lp); q2.resize((size_t)p.B * p.Lp);
  WM_HIP(hipMemcpyAsync(b2.data(), w.b2, b2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipMemcpyAsync(q2.data(), w.q2, q2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  if (!with_q) std::fill(q2.begin(), q2.end(), 1.0);
  return WM_OK;
}

// Q-free finalisation after jacobi_rows(with_q = false).  The rotated rows are b_i = s_i v_i^T
// up to the scale drift of ~3e4 float32 rotations per row (v_rsq_f32 rounds cos^2 + sin^2 a few
// 1e-8 below 1, every time); the DIRECTION v_i is good to the convergence threshold.  So the
// singular value is measured on the untouched input instead:  T = A0 B^T  (T[:, i] = A0 b_i^T =
// |b_i| s_i u_i), s_i = |T[:, i]| / |b_i| - the drift cancels, and T doubles as the left factor
// of the embed (u_i = T[:, i] / (|b_i| s_i)).  b2 = |b_i|^2, and q2 is returned as
// b2^2 / |T[:, i]|^2 so that sqrt(b2 / q2) is s_i like in the [A | I] formulation.
//   A0: dense [B][L][M] copy of the input rows;  T: dense [B][L][Lp] (left on the device).
// The finalisation's large products from split-f16 operands (k_hgemm, wm_ref_hier.inc) unless WM_RF_FINAL_F16=0 or the shapes
// do not allow 16-byte loads; the f32 k_sgemm otherwise.
static bool final_f16() {            // (read on every call, like WM_RF_HIER: tests switch it)
  const char* e = getenv("WM_RF_FINAL_F16");
  return !(e && atoi(e) == 0);
}

int fetch_norms_t(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const float* A0, float* T, std::vector<double>& b2,
                  std::vector<double>& q2, std::vector<unsigned char>* reliable = nullptr,
                  std::vector<double>* t2_raw = nullptr) {
  // all planes' products in one launch (grid.z = plane)
  hipLaunchKernelGGL(k_rf_rownorms, dim3(p.Lp, p.B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld, p.M, 0, w.b2, w.q2);
  if (final_f16() && hgemm_ok(A0, p.M, w.aug, p.ld)) {
    // A0 holds uint8 samples (every caller decomposes uint8 planes): exact in f16; B's rows are bounded by 255 sqrt(L)
    hipLaunchKernelGGL((k_hgemm<true, true, false>), dim3((p.Lp + 127) / 128, (p.L + 127) / 128, p.B), dim3(256), 0, ctx->stream, p.L, p.Lp, p.M,
                       A0, p.M, (size_t)p.L * p.M, w.aug, p.ld, p.aug_ps, 0, T, p.Lp, (size_t)p.L * p.Lp, (const float*)nullptr, (size_t)0, 1.0f);
  } else
  WM_TRY(sgemm_b(ctx, false, true, p.L, p.Lp, p.M, 1.0f, A0, p.M, (size_t)p.L * p.M, w.aug, p.ld, p.aug_ps, 0.0f, T, p.Lp,
                 (size_t)p.L * p.Lp, p.B));
  hipLaunchKernelGGL(k_rf_colnorms, dim3((p.Lp + 63) / 64, p.B), dim3(256), 0, ctx->stream, T, (size_t)p.L * p.Lp, p.L, p.Lp, w.q2);
  WM_HIP(hipGetLastError());
  b2.resize((size_t)p.B * p.Lp); q2.resize((size_t)p.B * p.Lp);
  WM_HIP(hipMemcpyAsync(b2.data(), w.b2, b2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipMemcpyAsync(q2.data(), w.q2, q2.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  if (t2_raw) *t2_raw = q2;                            // |T[:, i]|^2 as measured, before q2 becomes the sigma factor
  // |T[:, i]| = |b_i| s_i must agree with |b_i|^2 up to the rotations' scale drift (a few 1e-4).
  // A row that fails this is rounding residue of a rank-deficient plane - its direction is noise,
  // A0 b_i^T measures nothing - and keeps its own (tiny) norm as singular value.
  if (reliable) reliable->assign(b2.size(), 0);
  // a row below the numerical-null floor (the one the convergence test uses: NULL_ROW_RATIO |A|_F;
  // rotations preserve the Frobenius norm) whose A0 b_i^T is RESIDUE_RHO times larger than a singular
  // direction of that norm could give is rounding residue: its singular value is 0, as the float32
  // cast of LAPACK's 1e-13-sized values would be.  A genuinely small value (clean synthetic images)
  // has rho near 1 - off by c^2 (s_max / s_i)^2 / 2 at worst - and keeps its norm.
  std::vector<double> floor2(p.B, 0.0), bmax(p.B, 0.0);
  for (int z = 0; z < p.B; ++z) {
    double f2 = 0.0;
    for (int i = 0; i < p.Lp; ++i) { f2 += b2[(size_t)z * p.Lp + i]; bmax[z] = std::max(bmax[z], b2[(size_t)z * p.Lp + i]); }
    floor2[z] = NULL_ROW_RATIO * NULL_ROW_RATIO * f2;
  }
  // Scale drift of the rotated rows, measured: every row goes through the same ~3e4..3e5 rotations, each of which
  // shrinks it by the few 1e-8 that v_rsq_f32 leaves cos^2 + sin^2 below 1, so |b_i|^2 / s_i^2 is nearly the SAME
  // factor g < 1 for all rows of a plane (1 - g = 6e-4 at 1080p, 1.6e-3 at 8K).  Where both estimates exist (the
  // rows above the switch below) g_i = |b_i|^4 / |T[:, i]|^2 is known; its median calibrates the rows below the
  // switch, whose |b_i| is the only estimate: s_i = |b_i| / sqrt(g).  Without it the largest error of a whole
  // spectrum sat right below the switch (3.3e-6 s_1 at 4K, 7.8e-6 at 8K - tests/test_gpu_fullframe_large.py).
  static const bool drift_cal = !(getenv("WM_RF_DRIFT_CAL") && atoi(getenv("WM_RF_DRIFT_CAL")) == 0);
  std::vector<double> gcal(p.B, 1.0);
  {
    const double ratio0 = T_SWITCH * (double)ctx->ref_skip_thr;
    std::vector<double> g;
    for (int z = 0; z < p.B; ++z) {
      g.clear();
      for (int i = 0; i < p.Lp; ++i) {
        const size_t k = (size_t)z * p.Lp + i;
        if (!(q2[k] > 0.0 && b2[k] > 0.0) || b2[k] < ratio0 * ratio0 * bmax[z]) continue;
        const double gi = b2[k] * b2[k] / q2[k];
        if (fabs(sqrt(q2[k]) / b2[k] - 1.0) < DRIFT_TOL) g.push_back(gi);
      }
      if (g.size() >= 8) {
        std::sort(g.begin(), g.end());
        if (drift_cal) gcal[z] = g[g.size() / 2];
        if (getenv("WM_RF_DEBUG_DRIFT"))
          fprintf(stderr, "[wm_ref] plane %d: drift factor g over %zu rows: p05 %.3e  median %.3e  p95 %.3e  (1 - g)\n", z,
                  g.size(), 1.0 - g[g.size() / 20], 1.0 - g[g.size() / 2], 1.0 - g[g.size() - 1 - g.size() / 20]);
      }
    }
  }
  for (size_t i = 0; i < b2.size(); ++i) {
    const double t2 = q2[i];
    // |A0 b_i^T| / |b_i| agrees with |b_i| up to the rotations' scale drift unless b_i is residue
    const double rho = (t2 > 0.0 && b2[i] > 0.0) ? sqrt(t2) / b2[i] : 1e300;
    const bool consistent = fabs(rho - 1.0) < DRIFT_TOL;
    if (reliable && consistent) (*reliable)[i] = 1;
    // Which estimate: with a residual cosine c between rows i and j, |A0 b_i^T| / |b_i| is off by
    // c^2 (s_j / s_i)^2 / 2 relative, |b_i| by c^2 plus the scale drift (1e-4 relative, i.e. nothing in
    // absolute terms for a small value).  The Jacobi leaves c <= ref_skip_thr, so the first estimate is
    // used down to s_i = T_SWITCH * c * s_max (error <= 1.25e-5) and the plain norm below.
    const double ratio = T_SWITCH * (double)ctx->ref_skip_thr;
    if (consistent && b2[i] >= ratio * ratio * bmax[i / p.Lp]) {
      q2[i] = b2[i] * b2[i] / t2;                                       // sigma = |T[:, i]| / |b_i|
    } else if (rho > RESIDUE_RHO && b2[i] < floor2[i / p.Lp]) {
      b2[i] = 0.0; q2[i] = 1.0;                                         // residue below the floor: sigma = 0
    } else {
      q2[i] = gcal[i / p.Lp];                                           // sigma = |b_i| / sqrt(g): the norm, drift calibrated
    }
  }
  return WM_OK;
}

// copy the A part (first L rows, M columns) of every plane's Aug into a dense [B][L][M] array
int copy_a_part(wm_ctx* ctx, const RefPlan& p, const RefWs& w, float* dst) {
  for (int z = 0; z < p.B; ++z)
    WM_HIP(hipMemcpy2DAsync(dst + (size_t)z * p.L * p.M, (size_t)p.M * 4, w.aug + (size_t)z * p.aug_ps, (size_t)p.ld * 4,
                            (size_t)p.M * 4, p.L, hipMemcpyDeviceToDevice, ctx->stream));
  return WM_OK;
}

// sigma_i = |b_i| / |q_i| of one plane's Lp rows, sorted descending -> order, sig[L]
void sort_sigma(const RefPlan& p, const double* b2, const double* q2, std::vector<int>& order, std::vector<float>& sig) {
  std::vector<double> s(p.Lp);
  for (int i = 0; i < p.Lp; ++i) s[i] = sqrt(b2[i] / (q2[i] > 0 ? q2[i] : 1.0));
  order.resize(p.Lp);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return s[a] > s[b]; });
  sig.resize(p.L);
  for (int k = 0; k < p.L; ++k) sig[k] = (float)s[order[k]];
}

int check_ref_args(wm_ctx* ctx, const void* plane, int n_planes, int H, int W, int row_stride, size_t plane_stride) {
  WM_TRY(wmi::use_ctx(ctx));
  if (H <= 0 || W <= 0) return set_err(WM_ERR_BADARG, "H and W must be positive");
  if (n_planes <= 0 || n_planes > 65535) return set_err(WM_ERR_BADARG, "n_planes must be in 1..65535");
  if (!plane) return set_err(WM_ERR_BADARG, "plane pointer is NULL");
  if (row_stride < W) return set_err(WM_ERR_BADARG, "row_stride < W");
  if (n_planes > 1 && plane_stride < (size_t)row_stride * (H - 1) + W)
    return set_err(WM_ERR_BADARG, "plane_stride smaller than one plane");
  return WM_OK;
}

inline size_t span_of(int n_planes, int H, int W, int row_stride, size_t plane_stride) {
  return (size_t)(n_planes - 1) * plane_stride + (size_t)(H - 1) * row_stride + (size_t)W;
}


// ---------------------------------------------------------------------------
// cores: every plane-sized array is DEVICE memory; the small per-plane vectors (singular values, the
// embed coefficients) cross to the host once per call, where they are sorted / classified
// ---------------------------------------------------------------------------
struct RefSpectrum {                     // host-side result of one batched Jacobi + T = A0 B^T pass
  std::vector<double> b2, q2, t2;        // [B][Lp] |b_i|^2, the factor that turns it into sigma_i^2, |T[:, i]|^2
  std::vector<unsigned char> reliable;   // [B][Lp] u_i = T[:, i] / (|b_i| s_i) is a usable left vector
};

// planes (uint8, device) -> rotated rows B in w.aug, A0 copy in d_a0 [B][L][M], T = A0 B^T in d_t [B][L][Lp],
// spectrum on the host
int ref_decompose(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const uint8_t* d_planes, size_t row_stride,
                  size_t plane_stride, float* d_a0, float* d_t, JacobiUse use, RefSpectrum& sp) {
  hipLaunchKernelGGL((k_rf_load<uint8_t>), dim3(8, p.Lp, p.B), dim3(256), 0, ctx->stream, d_planes, row_stride,
                     plane_stride, p.transpose ? 1 : 0, w.aug, p.aug_ps, p.ld, p.L, p.Lp, p.M);
  WM_TRY(copy_a_part(ctx, p, w, d_a0));
  int sweeps = 0;
  WM_TRY(jacobi_rows(ctx, p, w, use, &sweeps));
  if (sweeps < 0) return set_err(WM_ERR_NOCONV, "SVD did not converge");
  WM_TRY(fetch_norms_t(ctx, p, w, d_a0, d_t, sp.b2, sp.q2, &sp.reliable, &sp.t2));
  return WM_OK;
}

// singular values of n planes on the device -> host sig [B][L]
int ref_sigma_core(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const uint8_t* d_planes, size_t row_stride,
                   size_t plane_stride, float* d_a0, float* d_t, float* sig_host) {
  RefSpectrum sp;
  WM_TRY(ref_decompose(ctx, p, w, d_planes, row_stride, plane_stride, d_a0, d_t, JR_SIGMA, sp));
  std::vector<int> order; std::vector<float> sig;
  for (int z = 0; z < p.B; ++z) {
    sort_sigma(p, &sp.b2[(size_t)z * p.Lp], &sp.q2[(size_t)z * p.Lp], order, sig);
    memcpy(sig_host + (size_t)z * p.L, sig.data(), (size_t)p.L * 4);
  }
  return WM_OK;
}

// Null-space completion of ONE rank-deficient plane z.  The reference injects alpha*Sw[k]
// for EVERY k < K (single:174-176): where the plane has no k-th singular direction LAPACK supplies some
// orthonormal completion of the null spaces.  Here: n deterministic pseudo-random vectors per side are
// projected off the valid singular vectors (twice), orthogonalised by the same block Jacobi, normalised,
// and  sum_k w_k u_k v_k^T  is added to Yw.  With every u and every v orthonormal,
// svd(Yw) = {s_i + w_i} U {w_k}: the reference's invariant  S(Cw)[:K] = Sc[:K] + alpha Sw[:K].
//   valid [Lp]: 1 for rows of B that are singular directions; wk [n]: the energies of the missing ranks;
//   tnorm2 [Lp]: |T[:, i]|^2.  d_yw: this plane's Yw in A layout [L][M].
int ref_complete_plane(wm_ctx* ctx, const RefPlan& p, const RefWs& w, int z, const std::vector<unsigned char>& valid,
                       const double* b2, const double* tnorm2, const std::vector<float>& wk, const float* d_t_z,
                       float* d_yw_z) {
  const int n = (int)wk.size();
  if (n == 0) return WM_OK;
  const float* Bz = w.aug + (size_t)z * p.aug_ps;
  // side 0: right vectors (length M, against the rows of B); side 1: left vectors (length L, against the columns of T)
  const RefPlan pv = make_plan(n, p.M), pu = make_plan(n, p.L);
  RefWs wv;
  // second workspace, planned for the larger of the two Jacobi runs (n x M; the n x L one has the same
  // block count and fewer columns and reuses its arrays).  tmp1: Zv [n][M] | Zu [n][L] | C [n][Lp] | coefficients [Lp]
  WM_TRY(plan_workspace(ctx, pv, wv, (size_t)n * (p.M + p.L + p.Lp) + p.Lp + 64, 16, 1));
  float* Zv = wv.tmp1; float* Zu = Zv + (size_t)n * p.M; float* C = Zu + (size_t)n * p.L; float* coef = C + (size_t)n * p.Lp;
  std::vector<float> gv(p.Lp), gu(p.Lp);
  for (int i = 0; i < p.Lp; ++i) {
    gv[i] = (valid[i] && b2[i] > 0.0) ? (float)(1.0 / b2[i]) : 0.0f;
    gu[i] = (valid[i] && tnorm2[i] > 0.0) ? (float)(1.0 / tnorm2[i]) : 0.0f;
  }
  hipLaunchKernelGGL(k_rf_pattern, dim3(8, n), dim3(256), 0, ctx->stream, Zv, p.M, 0x1234567u);
  hipLaunchKernelGGL(k_rf_pattern, dim3(8, n), dim3(256), 0, ctx->stream, Zu, p.L, 0x7654321u);
  for (int side = 0; side < 2; ++side) {
    float* Z = side == 0 ? Zv : Zu;
    WM_HIP(hipMemcpyAsync(coef, (side == 0 ? gv : gu).data(), (size_t)p.Lp * 4, hipMemcpyHostToDevice, ctx->stream));
    for (int pass = 0; pass < 2; ++pass) {          // project twice: classical Gram-Schmidt loses digits once
      if (side == 0) WM_TRY(sgemm(ctx, false, true, n, p.Lp, p.M, 1.0f, Z, p.M, Bz, p.ld, 0.0f, C, p.Lp));       // Z B^T
      else WM_TRY(sgemm(ctx, false, false, n, p.Lp, p.L, 1.0f, Z, p.L, d_t_z, p.Lp, 0.0f, C, p.Lp));              // Z T
      hipLaunchKernelGGL(k_rf_scale_cols_b, dim3(8, n, 1), dim3(256), 0, ctx->stream, C, (size_t)0, C, (size_t)0, p.Lp, coef, (size_t)0);
      if (side == 0) WM_TRY(sgemm(ctx, false, false, n, p.M, p.Lp, -1.0f, C, p.Lp, Bz, p.ld, 1.0f, Z, p.M));      // Z -= C B
      else WM_TRY(sgemm(ctx, false, true, n, p.L, p.Lp, -1.0f, C, p.Lp, d_t_z, p.Lp, 1.0f, Z, p.L));              // Z -= C T^T
    }
    WM_HIP(hipStreamSynchronize(ctx->stream));      // gv / gu are locals reused by the next side
  }
  // orthogonalise the rows of each Z with the block Jacobi (row span is preserved), then  Yw += Zu^T diag(d) Zv
  std::vector<double> nv, nu, dummy;
  std::vector<int> ordv, ordu;
  float* aug2 = wv.aug;
  auto orth = [&](const RefPlan& pp, float* Z, int len, std::vector<double>& norms2, std::vector<int>& ord) -> int {
    RefWs ww = wv;
    WM_TRY(get_pairs(ctx, pp, &ww.pairs));
    hipLaunchKernelGGL((k_rf_load<float>), dim3(8, pp.Lp, 1), dim3(256), 0, ctx->stream, Z, (size_t)len, (size_t)0, 0,
                       aug2, pp.aug_ps, pp.ld, pp.L, pp.Lp, pp.M);
    int sweeps = 0;
    WM_TRY(jacobi_rows(ctx, pp, ww, JR_ORTH, &sweeps));
    if (sweeps < 0) return set_err(WM_ERR_NOCONV, "null-space completion did not converge");
    WM_TRY(fetch_norms(ctx, pp, ww, false, norms2, dummy));
    ord.resize(pp.Lp);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return norms2[a] > norms2[b]; });
    // gather the n largest rows back into Z, normalised
    std::vector<float> sc(n);
    for (int k = 0; k < n; ++k) sc[k] = norms2[ord[k]] > 0.0 ? (float)(1.0 / sqrt(norms2[ord[k]])) : 0.0f;
    WM_HIP(hipMemcpyAsync(ww.order, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(ww.scale, sc.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rf_gather_rows, dim3(8, n), dim3(256), 0, ctx->stream, aug2, pp.ld, len, ww.order, ww.scale, Z, len);
    WM_HIP(hipStreamSynchronize(ctx->stream));       // ord / sc are locals
    return WM_OK;
  };
  // the workspace arrays of wv were sized for pv = (n, M); pu = (n, L) has the same row count and fewer columns
  WM_TRY(orth(pv, Zv, p.M, nv, ordv));
  WM_TRY(orth(pu, Zu, p.L, nu, ordu));
  // Zu rows scaled by the energies, then Yw += Zu^T Zv   ([L x n] [n x M])
  WM_HIP(hipMemcpyAsync(coef, wk.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rf_scale_rows, dim3(8, n, 1), dim3(256), 0, ctx->stream, Zu, (size_t)0, p.L, p.L, coef);
  WM_TRY(sgemm(ctx, true, false, p.L, p.M, n, 1.0f, Zu, p.L, Zv, p.M, 1.0f, d_yw_z, p.M));
  WM_HIP(hipStreamSynchronize(ctx->stream));         // wk is the caller's, coef reused
  return WM_OK;
}

// embed of p.B planes on the device.  d_in / d_out: uint8 planes (same strides; may alias);
// d_ywout: optional dense float [B][H][W]; sigma_w: HOST [B or 1][L]; sigma_c: HOST [B][L] out.
// d_yw [B][L][M] and d_t [B][L][Lp] are workspace.
int ref_embed_core(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const uint8_t* d_in, uint8_t* d_out, float* d_ywout,
                   size_t row_stride, size_t plane_stride, float* d_yw, float* d_t, const float* sigma_w,
                   size_t sigma_w_plane_stride, float* sigma_c, float alpha, int K, const int* sigma_w_ready = nullptr) {
  const size_t yw_ps = (size_t)p.L * p.M;
  RefSpectrum sp;
  // Yw starts as A itself (exactly the pixels): ref_decompose leaves that copy in d_yw
  WM_TRY(ref_decompose(ctx, p, w, d_in, row_stride, plane_stride, d_yw, d_t, JR_EMBED, sp));
  // sigma_w is first read here, after the host planes' decomposition (the long part of the call): a caller that is still
  // computing it - the watermark's own SVD on another context and thread, single:172-173's two independent statements -
  // says so with a flag: > 0 once sigma_w is written, < 0 if it never will be
  if (sigma_w_ready) {
    for (;;) {
      const int v = __atomic_load_n(sigma_w_ready, __ATOMIC_ACQUIRE);
      if (v > 0) break;
      if (v < 0) return set_err(WM_ERR_BADARG, "sigma_w was not produced (sigma_w_ready < 0)");
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }
  // U diag(alpha Sw) V^T = T diag(e) B  with  e_i = alpha * sw[rank(i)] / (s_i |b_i|^2), rank < K
  // (u_i = T[:, i] / (|b_i| s_i), v_i^T = b_i / |b_i|;  S_[:K] = Sc[:K] + alpha*Sw[:K]).
  // Split-f16 form of the product (k_hgemm): the factor e_i = ca_i cb_i goes half to each operand - B's rows to unit norm
  // (cb_i = 1 / |b_i|), T's columns to alpha sw (ca_i = alpha sw / (s_i |b_i|)) - so that both are inside f16's range whatever
  // the plane's scale; alpha * max(sw) must be (f32 product otherwise).
  bool f16_prod = final_f16() && hgemm_ok(d_t, p.Lp, w.aug, p.ld) && p.M % 4 == 0;
  for (int z = 0; z < p.B && f16_prod; ++z) {
    const float* sw = sigma_w + (size_t)z * sigma_w_plane_stride;
    for (int k = 0; k < std::min(K, p.L); ++k) if (!(fabs((double)alpha * sw[k]) < 3.0e4)) { f16_prod = false; break; }
  }
  std::vector<float> d((size_t)p.B * p.Lp * (f16_prod ? 2 : 1), 0.0f);       // row scales of B [, then column scales of T]
  std::vector<int> order; std::vector<float> sig;
  struct Todo { int z; std::vector<unsigned char> valid; std::vector<float> wk; std::vector<double> t2; };
  std::vector<Todo> todo;
  for (int z = 0; z < p.B; ++z) {
    const double* pb = &sp.b2[(size_t)z * p.Lp]; const double* pq = &sp.q2[(size_t)z * p.Lp];
    sort_sigma(p, pb, pq, order, sig);
    memcpy(sigma_c + (size_t)z * p.L, sig.data(), (size_t)p.L * 4);
    const float* sw = sigma_w + (size_t)z * sigma_w_plane_stride;
    const double s1 = sig.empty() ? 0.0 : (double)sig[0];
    Todo td; td.z = z; td.valid.assign(p.Lp, 0);
    td.t2.assign(sp.t2.begin() + (size_t)z * p.Lp, sp.t2.begin() + (size_t)(z + 1) * p.Lp);
    for (int i = 0; i < p.Lp; ++i) {
      // a singular direction: consistent T column, above the null ratio (sigma_i^2 = b2 / q2)
      const double s2 = pb[i] / (pq[i] > 0 ? pq[i] : 1.0);
      td.valid[i] = (sp.reliable[(size_t)z * p.Lp + i] && sqrt(s2) > NULL_RATIO * s1 && pb[i] > 0.0) ? 1 : 0;
    }
    for (int k = 0; k < std::min(K, p.L); ++k) {
      const int i = order[k];
      const double si = (double)sig[k];
      if (td.valid[i]) {
        if (f16_prod) {
          const double nb = sqrt(pb[i]);
          d[(size_t)z * p.Lp + i] = (float)(1.0 / nb);
          d[(size_t)(p.B + z) * p.Lp + i] = (float)((double)alpha * (double)sw[k] / (si * nb));
        } else d[(size_t)z * p.Lp + i] = (float)((double)alpha * (double)sw[k] / (si * pb[i]));
      } else td.wk.push_back(alpha * sw[k]);           // rank k has no singular direction in this plane: completed below
    }
    if (!td.wk.empty()) todo.push_back(std::move(td));
  }
  WM_HIP(hipMemcpyAsync(w.dvec, d.data(), d.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  // completion reads the unscaled rows of B: it runs before the rows are scaled in place
  if (!todo.empty()) {
    WM_HIP(hipStreamSynchronize(ctx->stream));
    for (const Todo& td : todo)
      WM_TRY(ref_complete_plane(ctx, p, w, td.z, td.valid, &sp.b2[(size_t)td.z * p.Lp], td.t2.data(), td.wk,
                                d_t + (size_t)td.z * p.L * p.Lp, d_yw + (size_t)td.z * yw_ps));
  }
  hipLaunchKernelGGL(k_rf_scale_rows, dim3(8, p.Lp, p.B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld, p.M, w.dvec);
  // Yw += T (diag(e) B):   [L x Lp] times [Lp x M], all planes in one launch
  if (f16_prod)
    hipLaunchKernelGGL((k_hgemm<false, false, true>), dim3((p.M + 127) / 128, (p.L + 127) / 128, p.B), dim3(256), 0, ctx->stream, p.L, p.M, p.Lp,
                       d_t, p.Lp, (size_t)p.L * p.Lp, w.aug, p.ld, p.aug_ps, 1, d_yw, p.M, yw_ps, w.dvec + (size_t)p.B * p.Lp, (size_t)p.Lp, 1.0f);
  else
  WM_TRY(sgemm_b(ctx, false, false, p.L, p.M, p.Lp, 1.0f, d_t, p.Lp, (size_t)p.L * p.Lp, w.aug, p.ld, p.aug_ps, 1.0f,
                 d_yw, p.M, yw_ps, p.B));
  hipLaunchKernelGGL(k_rf_quant, dim3(8, p.H, p.B), dim3(256), 0, ctx->stream, d_yw, yw_ps, p.M, p.transpose ? 1 : 0,
                     d_out, row_stride, plane_stride, d_ywout, p.H, p.W);
  WM_HIP(hipGetLastError());
  WM_HIP(hipStreamSynchronize(ctx->stream));         // d is a local
  return WM_OK;
}

// extract of B planes: sigma(stego) -> Sw_hat -> Uw[:L,:L] diag(Sw_hat) Vwt[:L,:L], zero-padded -> idct2.
// d_uw [H][L] and d_vwt [L][W] are the meta factors on the device (leading dimensions L and W);
// sigma_c HOST [B][L]; d_out dense float [B][H][W].  ws: tmp1 = Uw diag(sh) [B][L][L], tmp2 tail = GEMM intermediate [B][H][W].
// the product and the inverse DCT of single:214-218 for B planes with the estimates given: sh HOST [B][p.Lp] (entries at and
// beyond Lx are not read), Lx <= p.L the reference's truncation length - Uw[:Lx,:Lx] diag(sh) Vwt[:Lx,:Lx] lands in the
// top-left Lx x Lx corner of the zero plane (single:215-217), then idct2 of the whole H x W plane.
int ref_reconstruct_core(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const std::vector<float>& sh, const int Lx,
                         const float* d_uw, const float* d_vwt, float* d_us, float* d_mid, float* d_out) {
  const int L = p.L, H = p.H, W = p.W;
  float *dH, *dW;
  WM_TRY(get_dct_pair(ctx, H, W, &dH, &dW));
  if (final_f16() && L % 4 == 0 && W % 4 == 0 && H % 4 == 0 && hgemm_ok(d_uw, L, d_vwt, W) && hgemm_ok(d_mid, W, d_out, W)) {
    // Split-f16 products (k_hgemm): Uw, Vwt and the DCT bases are orthonormal; the estimates are scaled by a power of two so
    // that |X| <= max |sw_hat| * scale stays below 3e4, and the last product multiplies it out again (exact both ways).
    float mx = 0.0f;
    for (float v : sh) mx = fmaxf(mx, fabsf(v));
    int e = 0;
    if (mx > 0.0f && std::isfinite(mx)) { (void)frexpf(mx / 3.0e4f, &e); if (e < 0) e = 0; }
    const float scale = ldexpf(1.0f, -e);
    std::vector<float> shs(sh);
    for (float& v : shs) v *= scale;
    WM_HIP(hipMemcpyAsync(w.dvec, shs.data(), shs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemsetAsync(d_out, 0, (size_t)p.B * H * W * 4, ctx->stream));                                   // single:215
    const float* dHt = dH + (size_t)H * H;
    hipLaunchKernelGGL((k_hgemm<false, false, true>), dim3((Lx + 127) / 128, (Lx + 127) / 128, p.B), dim3(256), 0, ctx->stream, Lx, Lx, Lx,
                       d_uw, L, (size_t)0, d_vwt, W, (size_t)0, 0, d_out, W, (size_t)H * W, w.dvec, (size_t)p.Lp, 1.0f);          // single:214, 216-217
    hipLaunchKernelGGL((k_hgemm<false, false, false>), dim3((W + 127) / 128, (H + 127) / 128, p.B), dim3(256), 0, ctx->stream, H, W, H,
                       dHt, H, (size_t)0, d_out, W, (size_t)H * W, 0, d_mid, W, (size_t)H * W, (const float*)nullptr, (size_t)0, 1.0f);   // idct2: D_H^T X
    hipLaunchKernelGGL((k_hgemm<false, false, false>), dim3((W + 127) / 128, (H + 127) / 128, p.B), dim3(256), 0, ctx->stream, H, W, W,
                       d_mid, W, (size_t)H * W, dW, W, (size_t)0, 0, d_out, W, (size_t)H * W, (const float*)nullptr, (size_t)0, 1.0f / scale);   //        ... D_W   single:218
    WM_HIP(hipGetLastError());
    WM_HIP(hipStreamSynchronize(ctx->stream));       // (shs is a local)
    return WM_OK;
  }
  WM_HIP(hipMemcpyAsync(w.dvec, sh.data(), sh.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  // Uw[:Lx,:Lx] * sh (column scaling) per plane: the first Lx rows of Uw (leading dimension L)
  hipLaunchKernelGGL(k_rf_scale_cols_b, dim3(8, Lx, p.B), dim3(256), 0, ctx->stream, d_uw, (size_t)0, d_us, (size_t)L * L, L,
                     w.dvec, (size_t)p.Lp);
  WM_HIP(hipMemsetAsync(d_out, 0, (size_t)p.B * H * W * 4, ctx->stream));                                     // single:215
  WM_TRY(sgemm_b(ctx, false, false, Lx, Lx, Lx, 1.0f, d_us, L, (size_t)L * L, d_vwt, W, 0, 0.0f, d_out, W, (size_t)H * W, p.B));  // single:214, 216-217 (Vwt[:L,:L]: ld W)
  WM_TRY(sgemm_b(ctx, true, false, H, W, H, 1.0f, dH, H, 0, d_out, W, (size_t)H * W, 0.0f, d_mid, W, (size_t)H * W, p.B));   // idct2: D_H^T X
  WM_TRY(sgemm_b(ctx, false, false, H, W, W, 1.0f, d_mid, W, (size_t)H * W, dW, W, 0, 0.0f, d_out, W, (size_t)H * W, p.B));  //        ... D_W   single:218
  WM_HIP(hipStreamSynchronize(ctx->stream));         // sh may be a local of the caller
  return WM_OK;
}

int ref_extract_core(wm_ctx* ctx, const RefPlan& p, const RefWs& w, const float* s_cw, const float* sigma_c,
                     const float* d_uw, const float* d_vwt, float* d_us, float* d_mid, float* d_out, float alpha, int K) {
  const int L = p.L;
  const float a = fmaxf(alpha, 1e-8f);
  std::vector<float> sh((size_t)p.B * p.Lp, 0.0f);
  for (int z = 0; z < p.B; ++z)
    for (int i = 0; i < K; ++i)
      sh[(size_t)z * p.Lp + i] = (s_cw[(size_t)z * L + i] - sigma_c[(size_t)z * L + i]) / a;                  // single:212-213
  return ref_reconstruct_core(ctx, p, w, sh, L, d_uw, d_vwt, d_us, d_mid, d_out);
}

double nc_score(const float* sw, const float* scw, const float* sc, int L, float alpha) {
  // _nc(Sw[:L], (S_cw - Sc) / max(alpha, 1e-8))     single:297-301, 284-289
  const float a = fmaxf(alpha, 1e-8f);
  std::vector<double> x(L), y(L);
  double sa = 0, sb = 0;
  for (int i = 0; i < L; ++i) { x[i] = sw[i]; y[i] = (double)((scw[i] - sc[i]) / a); sa += x[i]; sb += y[i]; }
  sa /= L; sb /= L;
  double cov = 0, va = 0, vb = 0;
  for (int i = 0; i < L; ++i) { const double dx = x[i] - sa, dy = y[i] - sb; cov += dx * dy; va += dx * dx; vb += dy * dy; }
  return cov / (sqrt(va) * sqrt(vb) + 1e-8);
}

// small device -> host copy of a float vector (meta-sized arrays of the *_dev entry points)
int fetch_f32(wm_ctx* ctx, const float* d, size_t n, std::vector<float>& h) {
  h.resize(n);
  if (n) {
    WM_HIP(hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
  }
  return WM_OK;
}

}  // namespace

namespace wmi {
void hier_host_free(void* tab) { delete static_cast<HierTab*>(tab); }
}

// ===========================================================================
// C ABI (see include/wmhip.h)
// ===========================================================================
extern "C" {

// ---- device-pointer entry points: frames stay resident -------------------------------------
int wm_ref_sigma_planes_u8_dev(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H, int W,
                               int row_stride, size_t plane_stride) {
  WM_TRY(check_ref_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  if (!sigma) return set_err(WM_ERR_BADARG, "sigma is NULL");
  const RefPlan p = make_plan(H, W, n_planes);
  RefWs w;
  WM_TRY(plan_workspace(ctx, p, w, 16, (size_t)n_planes * p.L * (p.M + p.Lp)));   // tmp2: A0 [B][L][M] | T [B][L][Lp]
  std::vector<float> sig((size_t)n_planes * p.L);
  WM_TRY(ref_sigma_core(ctx, p, w, planes, (size_t)row_stride, plane_stride, w.tmp2, w.tmp2 + (size_t)n_planes * p.L * p.M,
                        sig.data()));
  WM_HIP(hipMemcpyAsync(sigma, sig.data(), sig.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_embed_planes_u8_dev(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego, float* sigma_c,
                               float* yw, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                               size_t sigma_w_plane_stride, float alpha, int K) {
  WM_TRY(check_ref_args(ctx, host, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_w || !stego || !sigma_c) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, n_planes);
  if (K < 0 || K > p.L) return set_err(WM_ERR_BADARG, "K must be in 0..min(H,W)");
  RefWs w;
  WM_TRY(plan_workspace(ctx, p, w, 16, (size_t)n_planes * p.L * (p.M + p.Lp)));   // tmp2: Yw [B][L][M] | T [B][L][Lp]
  std::vector<float> sw, sc((size_t)n_planes * p.L);
  WM_TRY(fetch_f32(ctx, sigma_w, sigma_w_plane_stride ? (size_t)(n_planes - 1) * sigma_w_plane_stride + p.L : (size_t)p.L, sw));
  // pixels the tiles do not cover do not exist in this mode: every pixel of stego is written by the quantiser
  WM_TRY(ref_embed_core(ctx, p, w, host, stego, yw, (size_t)row_stride, plane_stride, w.tmp2,
                        w.tmp2 + (size_t)n_planes * p.L * p.M, sw.data(), sigma_w_plane_stride, sc.data(), alpha, K));
  WM_HIP(hipMemcpyAsync(sigma_c, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_extract_planes_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                                 const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                                 size_t plane_stride, float alpha, int K) {
  WM_TRY(check_ref_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_c || !Uw || !Vwt || !out) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, n_planes);
  const int L = p.L;
  if (K < 0 || K > L) return set_err(WM_ERR_BADARG, "K must be in 0..min(H,W)");
  RefWs w;
  // tmp1: Uw diag(sh) [B][L][L];  tmp2: A0 [B][L][M] | T [B][L][Lp], reused afterwards as the GEMM intermediate [B][H][W]
  WM_TRY(plan_workspace(ctx, p, w, (size_t)n_planes * L * L + 16,
                        std::max((size_t)n_planes * p.L * (p.M + p.Lp), (size_t)n_planes * H * W)));
  std::vector<float> s_cw((size_t)n_planes * L), sc;
  WM_TRY(ref_sigma_core(ctx, p, w, stego, (size_t)row_stride, plane_stride, w.tmp2, w.tmp2 + (size_t)n_planes * p.L * p.M,
                        s_cw.data()));                                                                   // single:205
  WM_TRY(fetch_f32(ctx, sigma_c, (size_t)n_planes * L, sc));
  return ref_extract_core(ctx, p, w, s_cw.data(), sc.data(), Uw, Vwt, w.tmp1, w.tmp2, out, alpha, K);
}

int wm_ref_detect_planes_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                                double* scores, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                                float alpha) {
  WM_TRY(check_ref_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_c || !sigma_w || !scores) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, n_planes);
  RefWs w;
  WM_TRY(plan_workspace(ctx, p, w, 16, (size_t)n_planes * p.L * (p.M + p.Lp)));
  std::vector<float> s_cw((size_t)n_planes * p.L), sc, sw;
  WM_TRY(ref_sigma_core(ctx, p, w, stego, (size_t)row_stride, plane_stride, w.tmp2, w.tmp2 + (size_t)n_planes * p.L * p.M,
                        s_cw.data()));
  WM_TRY(fetch_f32(ctx, sigma_c, (size_t)n_planes * p.L, sc));
  WM_TRY(fetch_f32(ctx, sigma_w, (size_t)p.L, sw));
  std::vector<double> sco(n_planes);
  for (int z = 0; z < n_planes; ++z)
    sco[z] = nc_score(sw.data(), &s_cw[(size_t)z * p.L], &sc[(size_t)z * p.L], p.L, alpha);
  WM_HIP(hipMemcpyAsync(scores, sco.data(), sco.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

// ---- host-pointer entry points ------------------------------------------------------------------
int wm_ref_sigma_planes_u8(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H, int W,
                           int row_stride, size_t plane_stride) {
  WM_TRY(check_ref_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  if (!sigma) return set_err(WM_ERR_BADARG, "sigma is NULL");
  const RefPlan p = make_plan(H, W, n_planes);
  RefWs w;
  const size_t n_in = span_of(n_planes, H, W, row_stride, plane_stride);
  // tmp1: uint8 input span; tmp2: A0 [B][L][M] | T [B][L][Lp]
  WM_TRY(plan_workspace(ctx, p, w, (n_in + 3) / 4 + 4, (size_t)n_planes * p.L * (p.M + p.Lp)));
  uint8_t* d_in = (uint8_t*)w.tmp1;
  WM_HIP(hipMemcpyAsync(d_in, planes, n_in, hipMemcpyHostToDevice, ctx->stream));
  return ref_sigma_core(ctx, p, w, d_in, (size_t)row_stride, plane_stride, w.tmp2, w.tmp2 + (size_t)n_planes * p.L * p.M, sigma);
}

int wm_ref_sigma_u8(wm_ctx* ctx, const uint8_t* plane, float* sigma, int H, int W, int row_stride) {
  return wm_ref_sigma_planes_u8(ctx, plane, sigma, 1, H, W, row_stride, (size_t)H * row_stride);
}

int wm_ref_embed_planes_u8_when(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, const int* sigma_w_ready,
                                uint8_t* stego, float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                                size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K) {
  WM_TRY(check_ref_args(ctx, host, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_w || !stego || !sigma_c) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, n_planes);
  if (K < 0 || K > p.L) return set_err(WM_ERR_BADARG, "K must be in 0..min(H,W)");
  RefWs w;
  const size_t n_in = span_of(n_planes, H, W, row_stride, plane_stride);
  const size_t n_in16 = (n_in + 15) & ~(size_t)15;
  const size_t yw_elems = yw ? (size_t)n_planes * H * W : 0;
  // tmp1: uint8 input + output spans (+ dense float Yw for the caller);
  // tmp2: Yw in A layout [B][L][M] (starts as A0, the pixels) | T [B][L][Lp]
  WM_TRY(plan_workspace(ctx, p, w, (2 * n_in16) / 4 + 8 + yw_elems, (size_t)n_planes * p.L * (p.M + p.Lp)));
  uint8_t* d_in = (uint8_t*)w.tmp1;
  uint8_t* d_out = d_in + n_in16;
  float* d_ywout = yw ? (float*)(d_out + n_in16) : nullptr;
  WM_HIP(hipMemcpyAsync(d_in, host, n_in, hipMemcpyHostToDevice, ctx->stream));
  // bytes between rows / planes that belong to the caller's stego buffer are carried through
  if (stego != host) WM_HIP(hipMemcpyAsync(d_out, stego, n_in, hipMemcpyHostToDevice, ctx->stream));
  else WM_HIP(hipMemcpyAsync(d_out, d_in, n_in, hipMemcpyDeviceToDevice, ctx->stream));
  WM_TRY(ref_embed_core(ctx, p, w, d_in, d_out, d_ywout, (size_t)row_stride, plane_stride, w.tmp2,
                        w.tmp2 + (size_t)n_planes * p.L * p.M, sigma_w, sigma_w_plane_stride, sigma_c, alpha, K, sigma_w_ready));
  WM_HIP(hipMemcpyAsync(stego, d_out, n_in, hipMemcpyDeviceToHost, ctx->stream));
  if (yw) WM_HIP(hipMemcpyAsync(yw, d_ywout, yw_elems * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_embed_planes_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego, float* sigma_c,
                           float* yw, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                           size_t sigma_w_plane_stride, float alpha, int K) {
  return wm_ref_embed_planes_u8_when(ctx, host, sigma_w, nullptr, stego, sigma_c, yw, n_planes, H, W, row_stride, plane_stride,
                                     sigma_w_plane_stride, alpha, K);
}

int wm_ref_embed_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego, float* sigma_c,
                    float* yw, int H, int W, int row_stride, float alpha, int K) {
  return wm_ref_embed_planes_u8(ctx, host, sigma_w, stego, sigma_c, yw, 1, H, W, row_stride,
                                (size_t)H * row_stride, 0, alpha, K);
}

// thin SVD of dct2(plane) (apply_dct != 0) or of the plane itself: U [H x L], S [L], Vt [L x W]
// Watermark-side SVD of n_planes planes (the B, G, R planes of a colour watermark, single:128-134) in ONE batch: a single
// full-frame SVD is latency-bound on a fraction of the chip, three planes in every launch cost 1.4 x one.
//   planes [n][H][row_stride..] float32 host, U [n][H][L], S [n][L], Vt [n][L][W] host
int wm_ref_svd_planes_f32(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt, int n_planes, int H, int W,
                          int row_stride, size_t plane_stride, int apply_dct) {
  WM_TRY(check_ref_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  if (!U || !S || !Vt) return set_err(WM_ERR_BADARG, "U/S/Vt is NULL");
  const int B = n_planes;
  const RefPlan p = make_plan(H, W, B);
  RefWs w;
  const size_t n_in = span_of(B, H, W, row_stride, plane_stride);
  const size_t n_in_a = (n_in + 63) & ~(size_t)63, hw = (size_t)H * W, fl = (size_t)p.L * (p.M + p.L), lp64 = ((size_t)p.Lp + 63) & ~(size_t)63;
  // tmp1: input planes | DCT planes [B][H][W] | every plane's sorted factors in the caller's layout, U [H][L] | Vt [L][W] | per plane: order, 1 / |q_i|,
  // 1 / |b_i|;  tmp2: DCT intermediate [B][H][W], then T = A0 B^T [B][L][Lp]
  WM_TRY(plan_workspace(ctx, p, w, n_in_a + (size_t)B * hw + (size_t)B * fl + 3 * (size_t)B * lp64, std::max((size_t)B * hw, (size_t)B * p.L * p.Lp)));
  float* d_in = w.tmp1;
  float* d_c = d_in + n_in_a;            // [B][H][W]
  float* d_f = d_c + (size_t)B * hw;     // [B]( U [H][L] | Vt [L][W] )
  int* d_ord = (int*)(d_f + (size_t)B * fl);
  float* d_sq = (float*)(d_ord + (size_t)B * lp64);
  float* d_sb = d_sq + (size_t)B * lp64;
  WM_HIP(hipMemcpyAsync(d_in, planes, n_in * 4, hipMemcpyHostToDevice, ctx->stream));
  const float* src = d_in; size_t src_stride = (size_t)row_stride, src_ps = plane_stride;
  if (apply_dct) {
    float *dH, *dW;
    WM_TRY(get_dct_pair(ctx, H, W, &dH, &dW));
    WM_TRY(sgemm_b(ctx, false, false, H, W, H, 1.0f, dH, H, 0, d_in, row_stride, plane_stride, 0.0f, w.tmp2, W, hw, B));   // D_H X
    WM_TRY(sgemm_b(ctx, false, true, H, W, W, 1.0f, w.tmp2, W, hw, dW, W, 0, 0.0f, d_c, W, hw, B));                      // (D_H X) D_W^T
    src = d_c; src_stride = (size_t)W; src_ps = hw;
  }
  hipLaunchKernelGGL((k_rf_load<float>), dim3(8, p.Lp, B), dim3(256), 0, ctx->stream, src, src_stride, src_ps,
                     p.transpose ? 1 : 0, w.aug, p.aug_ps, p.ld, p.L, p.Lp, p.M);
  int sweeps = 0;
  WM_TRY(jacobi_rows(ctx, p, w, JR_SVD, &sweeps));
  if (sweeps < 0) return set_err(WM_ERR_NOCONV, "SVD did not converge");
  std::vector<double> b2, q2;
  WM_TRY(fetch_norms(ctx, p, w, true, b2, q2));
  // Singular values.  |b_i| / |q_i| is exact as long as B = Qt A holds, but the two parts of a row round
  // independently through ~1e5 MFMA row updates: measured 7e-5 relative low at 8K (3e-5 at 1080p), the same for
  // every value.  Like the sigma-only path (fetch_norms_t) the large values are therefore measured on the untouched
  // input, s_i = |A0 b_i^T| / |b_i| (error c^2 (s_max / s_i)^2 / 2 with the residual cosine c <= ref_skip_thr), and
  // the ratio of the two estimates there (median) calibrates |b_i| / |q_i| for the values below the switch.
  float* d_T = w.tmp2;
  WM_TRY(sgemm_b(ctx, p.transpose, true, p.L, p.Lp, p.M, 1.0f, src, (int)src_stride, src_ps, w.aug, p.ld, p.aug_ps, 0.0f, d_T, p.Lp,
                 (size_t)p.L * p.Lp, B));
  hipLaunchKernelGGL(k_rf_colnorms, dim3((p.Lp + 63) / 64, B), dim3(256), 0, ctx->stream, d_T, (size_t)p.L * p.Lp, p.L, p.Lp, w.q2);
  WM_HIP(hipGetLastError());
  std::vector<double> t2all((size_t)B * p.Lp);
  WM_HIP(hipMemcpyAsync(t2all.data(), w.q2, t2all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<float> sq((size_t)B * lp64, 0.0f), sb((size_t)B * lp64, 0.0f), sig;
  std::vector<int> order, ord_all((size_t)B * lp64, 0);
  for (int z = 0; z < B; ++z) {
    const double* b2z = &b2[(size_t)z * p.Lp]; const double* q2z = &q2[(size_t)z * p.Lp]; const double* t2 = &t2all[(size_t)z * p.Lp];
    std::vector<double> q2s(q2z, q2z + p.Lp);           // q2 itself still normalises the columns of the short-side factor
    double smax2 = 0.0;
    for (int i = 0; i < p.Lp; ++i) if (q2z[i] > 0.0) smax2 = std::max(smax2, b2z[i] / q2z[i]);
    const double ratio = T_SWITCH * (double)ctx->ref_skip_thr;
    std::vector<double> g;
    std::vector<unsigned char> above(p.Lp, 0);
    for (int i = 0; i < p.Lp; ++i) {
      if (!(b2z[i] > 0.0 && q2z[i] > 0.0 && t2[i] > 0.0)) continue;
      const double s2 = b2z[i] / q2z[i];
      const double rho = sqrt(t2[i] * q2z[i]) / b2z[i];            // |T[:, i]| against |b_i| * (|b_i| / |q_i|)
      if (fabs(rho - 1.0) < DRIFT_TOL && s2 >= ratio * ratio * smax2) { above[i] = 1; g.push_back(b2z[i] * b2z[i] / (q2z[i] * t2[i])); }
    }
    double gcal = 1.0;
    if (g.size() >= 8) { std::sort(g.begin(), g.end()); gcal = g[g.size() / 2]; }
    for (int i = 0; i < p.Lp; ++i)
      q2s[i] = above[i] ? b2z[i] * b2z[i] / t2[i] : q2z[i] * gcal;
    sort_sigma(p, b2z, q2s.data(), order, sig);
    memcpy(S + (size_t)z * p.L, sig.data(), (size_t)p.L * 4);
    // short-side factor: columns q_i/|q_i|   (rows of Qt), long-side factor: rows b_i/|b_i|
    for (int k = 0; k < p.L; ++k) {
      const int i = order[k];
      ord_all[(size_t)z * lp64 + k] = i;
      sq[(size_t)z * lp64 + k] = q2z[i] > 0 ? (float)(1.0 / sqrt(q2z[i])) : 0.0f;
      sb[(size_t)z * lp64 + k] = b2z[i] > 0 ? (float)(1.0 / sqrt(b2z[i])) : 0.0f;
    }
  }
  WM_HIP(hipMemcpyAsync(d_ord, ord_all.data(), ord_all.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipMemcpyAsync(d_sq, sq.data(), sq.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipMemcpyAsync(d_sb, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  // The factors leave the device in the caller's layout (round 3 downloaded both sorted factors row-major and transposed one
  // of them on the host, element by element: 5 - 10 ms of a 1080p call, ~0.3 s of an 8K one).
  //   A = X   (H <= W): U[r][k] = q_k[r] / |q_k|  (transposing gather of the Qt part),  Vt[k][c] = b_k[c] / |b_k|  (straight)
  //   A = X^T (H >  W): U[r][k] = b_k[r] / |b_k|  (transposing gather of the B part),   Vt[k][c] = q_k[c] / |q_k|  (straight)
  float* d_u = d_f; float* d_vt = d_f + (size_t)H * p.L;
  const int lp = (int)lp64;
  if (!p.transpose) {
    hipLaunchKernelGGL(k_rf_gather_rows_t, dim3((p.L + 31) / 32, (p.L + 31) / 32, B), dim3(256), 0, ctx->stream, w.aug + p.M, p.aug_ps, p.ld,
                       p.L, p.L, d_ord, d_sq, lp, d_u, fl, p.L);
    hipLaunchKernelGGL(k_rf_gather_rows_b, dim3(8, p.L, B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld, p.M, d_ord, d_sb, lp, d_vt, fl, W);
  } else {
    hipLaunchKernelGGL(k_rf_gather_rows_t, dim3((p.M + 31) / 32, (p.L + 31) / 32, B), dim3(256), 0, ctx->stream, w.aug, p.aug_ps, p.ld,
                       p.M, p.L, d_ord, d_sb, lp, d_u, fl, p.L);
    hipLaunchKernelGGL(k_rf_gather_rows_b, dim3(8, p.L, B), dim3(256), 0, ctx->stream, w.aug + p.M, p.aug_ps, p.ld, p.L, d_ord, d_sq, lp, d_vt, fl, W);
  }
  WM_HIP(hipGetLastError());
  for (int z = 0; z < B; ++z) {
    WM_HIP(hipMemcpyAsync(U + (size_t)z * H * p.L, d_u + (size_t)z * fl, (size_t)H * p.L * 4, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipMemcpyAsync(Vt + (size_t)z * p.L * W, d_vt + (size_t)z * fl, (size_t)p.L * W * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_svd_f32(wm_ctx* ctx, const float* plane, float* U, float* S, float* Vt, int H, int W,
                   int row_stride, int apply_dct) {
  return wm_ref_svd_planes_f32(ctx, plane, U, S, Vt, 1, H, W, row_stride, (size_t)H * row_stride, apply_dct);
}


// extract: n_planes stego planes share one watermark decomposition (frames of a video): their SVDs
// run batched, the three GEMMs per plane run as one launch each (grid.z = plane).
int wm_ref_extract_planes_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                             const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                             size_t plane_stride, float alpha, int K) {
  WM_TRY(check_ref_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_c || !Uw || !Vwt || !out) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, n_planes);
  const int L = p.L;
  if (K < 0 || K > L) return set_err(WM_ERR_BADARG, "K must be in 0..min(H,W)");
  RefWs w;
  const size_t n_in = span_of(n_planes, H, W, row_stride, plane_stride);
  const size_t n_in4 = (n_in + 3) / 4 + 4;
  // tmp1: uint8 stego span | Uw[:L,:L] | Vwt [L][W] | Uw diag(sh) [B][L][L] | result [B][H][W]
  // tmp2: A0 [B][L][M] | T [B][L][Lp], reused afterwards as the GEMM intermediate [B][H][W]
  WM_TRY(plan_workspace(ctx, p, w, n_in4 + (size_t)L * L + (size_t)L * W + (size_t)n_planes * L * L + (size_t)n_planes * H * W,
                        std::max((size_t)n_planes * p.L * (p.M + p.Lp), (size_t)n_planes * H * W)));
  uint8_t* d_in = (uint8_t*)w.tmp1;
  float* d_u = w.tmp1 + n_in4; float* d_v = d_u + (size_t)L * L; float* d_us = d_v + (size_t)L * W;
  float* d_full = d_us + (size_t)n_planes * L * L;
  WM_HIP(hipMemcpyAsync(d_in, stego, n_in, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipMemcpyAsync(d_u, Uw, (size_t)L * L * 4, hipMemcpyHostToDevice, ctx->stream));            // Uw[:L,:L] (H x L, rows < L)
  WM_HIP(hipMemcpyAsync(d_v, Vwt, (size_t)L * W * 4, hipMemcpyHostToDevice, ctx->stream));           // Vwt [L][W]; [:L,:L] = leading dimension W
  std::vector<float> s_cw((size_t)n_planes * L);
  WM_TRY(ref_sigma_core(ctx, p, w, d_in, (size_t)row_stride, plane_stride, w.tmp2, w.tmp2 + (size_t)n_planes * p.L * p.M,
                        s_cw.data()));                                                                   // single:205
  WM_TRY(ref_extract_core(ctx, p, w, s_cw.data(), sigma_c, d_u, d_v, d_us, w.tmp2, d_full, alpha, K));
  WM_HIP(hipMemcpyAsync(out, d_full, (size_t)n_planes * H * W * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_extract_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw, const float* Vwt,
                      float* out, int H, int W, int row_stride, float alpha, int K) {
  return wm_ref_extract_planes_u8(ctx, stego, sigma_c, Uw, Vwt, out, 1, H, W, row_stride, (size_t)H * row_stride,
                                  alpha, K);
}

// single:214-218 on its own, with the estimates given: the drop-in uses it when the stego's size is not the meta's (a
// resized or cropped stego - the reference goes on with L = the shortest of the four lengths, single:210)
int wm_ref_reconstruct_f32(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt, float* out, int H, int W,
                           int L) {
  WM_TRY(check_ref_args(ctx, Uw, 1, H, W, W, (size_t)H * W));
  if (!sw_hat || !Vwt || !out) return set_err(WM_ERR_BADARG, "NULL argument");
  const RefPlan p = make_plan(H, W, 1);
  const int Lm = p.L;
  if (L < 0 || L > Lm) return set_err(WM_ERR_BADARG, "L must be in 0..min(H,W)");
  RefWs w;
  // tmp1: Uw[:Lm,:Lm] | Vwt [Lm][W] | Uw diag(sh) [Lm][Lm] | result [H][W];  tmp2: GEMM intermediate [H][W]
  WM_TRY(plan_workspace(ctx, p, w, 2 * (size_t)Lm * Lm + (size_t)Lm * W + (size_t)H * W + 16, (size_t)H * W));
  float* d_u = w.tmp1; float* d_v = d_u + (size_t)Lm * Lm; float* d_us = d_v + (size_t)Lm * W;
  float* d_full = d_us + (size_t)Lm * Lm;
  WM_HIP(hipMemcpyAsync(d_u, Uw, (size_t)Lm * Lm * 4, hipMemcpyHostToDevice, ctx->stream));          // Uw[:Lm,:Lm] (H x Lm, rows < Lm)
  WM_HIP(hipMemcpyAsync(d_v, Vwt, (size_t)Lm * W * 4, hipMemcpyHostToDevice, ctx->stream));
  std::vector<float> sh((size_t)p.Lp, 0.0f);
  for (int i = 0; i < L; ++i) sh[i] = sw_hat[i];
  if (L == 0) {
    WM_HIP(hipMemsetAsync(d_full, 0, (size_t)H * W * 4, ctx->stream));       // idct2 of the zero plane
  } else {
    WM_TRY(ref_reconstruct_core(ctx, p, w, sh, L, d_u, d_v, d_us, w.tmp2, d_full));
  }
  WM_HIP(hipMemcpyAsync(out, d_full, (size_t)H * W * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_ref_last_flops(wm_ctx* ctx, double* flops_out, int* two_level_out) {
  if (!ctx || !flops_out || !two_level_out) return set_err(WM_ERR_BADARG, "NULL argument");
  *flops_out = ctx->ref_last_flops; *two_level_out = ctx->ref_last_hier;
  return WM_OK;
}

// outer sweeps the last full-frame SVD on this context took (for flop accounting)
int wm_ref_last_sweeps(wm_ctx* ctx, int* sweeps_out) {
  if (!ctx || !sweeps_out) return set_err(WM_ERR_BADARG, "NULL argument");
  *sweeps_out = ctx->ref_last_sweeps;
  return WM_OK;
}

// detect over n_planes stego planes of one watermark (frames of a clip): batched SVDs, NC per plane
int wm_ref_detect_planes_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                            double* scores, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                            float alpha) {
  WM_TRY(check_ref_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!sigma_c || !sigma_w || !scores) return set_err(WM_ERR_BADARG, "NULL argument");
  const int L = std::min(H, W);
  std::vector<float> s_cw((size_t)n_planes * L);
  WM_TRY(wm_ref_sigma_planes_u8(ctx, stego, s_cw.data(), n_planes, H, W, row_stride, plane_stride));
  for (int z = 0; z < n_planes; ++z)
    scores[z] = nc_score(sigma_w, &s_cw[(size_t)z * L], sigma_c + (size_t)z * L, L, alpha);
  return WM_OK;
}

int wm_ref_detect_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                     double* score, int H, int W, int row_stride, float alpha) {
  return wm_ref_detect_planes_u8(ctx, stego, sigma_c, sigma_w, score, 1, H, W, row_stride, (size_t)H * row_stride, alpha);
}

}  // extern "C"
