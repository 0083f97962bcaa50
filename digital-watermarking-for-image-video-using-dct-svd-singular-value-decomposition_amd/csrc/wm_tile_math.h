// Per-tile arithmetic of the DCT-SVD watermark hot path: one 8x8 tile per
// GPU lane, everything in registers (no LDS, no cross-lane traffic).
//
// Reference statements this replaces (citations into /root/reference,
// app_dct_svd_single.py = "single"), applied to one 8x8 tile instead of the
// whole plane (tile-mode):
//   a2  dct2            single:32-33   -> dct8x8()
//   a3  np.linalg.svd   single:172-173 -> jacobi_svd8()   (one-sided Jacobi)
//   a4  S_[:K] += a*Sw  single:174-175 -> embed_tile()
//   a5  U diag(S_) Vt   single:176     -> embed_tile()
//   a6  idct2           single:35-36   -> idct8x8()
//   a7  clip + astype   single:27      -> quant_u8()       (truncation)
//   a8  (S_cw-Sc)/alpha single:212-213 -> extract_tile()
//   a9  Uw diag() Vwt   single:214-218 -> extract_tile()
//
// The header is host/device: hipcc compiles it into the gfx950 kernels
// (wm_kernels.hip); tests compile the very same functions with g++ to check
// the arithmetic on the CPU (tests/host_harness.cpp).  The host build is a
// test harness, never a product path.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define WM_HD __host__ __device__ __forceinline__
#else
#define WM_HD inline
#endif

namespace wm {

// ---- fast reciprocal / sqrt (v_rcp_f32, v_rsq_f32, v_sqrt_f32: 1 ulp) -------
WM_HD float frcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
WM_HD float frsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
WM_HD float fsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
WM_HD float ffma(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmaf(a, b, c);
#else
  return fmaf(a, b, c);
#endif
}
// wave-wide OR of a per-lane predicate (host: identity)
WM_HD bool wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(p) != 0ull;
#else
  return p;
#endif
}

// ---- 8-point orthonormal DCT-II constants: D[k][m] = ck*cos((2m+1)k*pi/16) --
// c_k = 0.5*cos(k*pi/16) for k>=1; row 0 is sqrt(1/8) = C4.
constexpr float C1 = 0.49039264020161522f;
constexpr float C2 = 0.46193976625564337f;
constexpr float C3 = 0.41573480615127262f;
constexpr float C4 = 0.35355339059327379f;
constexpr float C5 = 0.27778511650980114f;
constexpr float C6 = 0.19134171618254492f;
constexpr float C7 = 0.09754516100806417f;

// forward 1-D DCT-II of 8 samples, even/odd split (36 flop-instr)
WM_HD void dct8(float& x0, float& x1, float& x2, float& x3,
                float& x4, float& x5, float& x6, float& x7) {
  const float s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
  const float d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
  const float e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
  x0 = C4 * (e0 + e1);
  x4 = C4 * (e0 - e1);
  x2 = ffma(C2, e2, C6 * e3);
  x6 = ffma(C6, e2, -C2 * e3);
  x1 = ffma(C1, d0, ffma(C3, d1, ffma(C5, d2, C7 * d3)));
  x3 = ffma(C3, d0, ffma(-C7, d1, ffma(-C1, d2, -C5 * d3)));
  x5 = ffma(C5, d0, ffma(-C1, d1, ffma(C7, d2, C3 * d3)));
  x7 = ffma(C7, d0, ffma(-C5, d1, ffma(C3, d2, -C1 * d3)));
}

// inverse (DCT-III with the same orthonormal scaling)
WM_HD void idct8(float& x0, float& x1, float& x2, float& x3,
                 float& x4, float& x5, float& x6, float& x7) {
  const float p = C4 * (x0 + x4), q = C4 * (x0 - x4);
  const float r = ffma(C2, x2, C6 * x6), t = ffma(C6, x2, -C2 * x6);
  const float e0 = p + r, e3 = p - r, e1 = q + t, e2 = q - t;
  const float o0 = ffma(C1, x1, ffma(C3, x3, ffma(C5, x5, C7 * x7)));
  const float o1 = ffma(C3, x1, ffma(-C7, x3, ffma(-C1, x5, -C5 * x7)));
  const float o2 = ffma(C5, x1, ffma(-C1, x3, ffma(C7, x5, C3 * x7)));
  const float o3 = ffma(C7, x1, ffma(-C5, x3, ffma(C3, x5, -C1 * x7)));
  x0 = e0 + o0; x7 = e0 - o0;
  x1 = e1 + o1; x6 = e1 - o1;
  x2 = e2 + o2; x5 = e2 - o2;
  x3 = e3 + o3; x4 = e3 - o3;
}

// 2-D transforms on a[row][col]:  C = D X D^T   /   X = D^T C D
WM_HD void dct8x8(float (&a)[8][8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
    dct8(a[0][c], a[1][c], a[2][c], a[3][c], a[4][c], a[5][c], a[6][c], a[7][c]);
#pragma unroll
  for (int r = 0; r < 8; ++r)
    dct8(a[r][0], a[r][1], a[r][2], a[r][3], a[r][4], a[r][5], a[r][6], a[r][7]);
}
WM_HD void idct8x8(float (&a)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
    idct8(a[r][0], a[r][1], a[r][2], a[r][3], a[r][4], a[r][5], a[r][6], a[r][7]);
#pragma unroll
  for (int c = 0; c < 8; ++c)
    idct8(a[0][c], a[1][c], a[2][c], a[3][c], a[4][c], a[5][c], a[6][c], a[7][c]);
}

// ---- one-sided (Hestenes) Jacobi SVD of an 8x8 matrix ------------------------
// Row-cyclic pair order with de Rijk's norm ordering (the larger column of a
// pair always ends in the lower index), so singular values come out sorted
// descending like LAPACK's.  On return
//   a  = B = A*V     (mutually orthogonal columns, |b_i| descending)
//   v  = V           (accumulated plane rotations; WITH_V only)
//   n2 = |b_i|^2 ,  vn2 = |v_i|^2 (exactly 1 in exact arithmetic; carried so the
//        1-ulp drift of v_rsq_f32 cancels in sigma_i = |b_i| / |v_i|).
// Sweeps stop when a whole sweep saw no pair with cos^2 > CONV2 for any tile
// of the wave (Jacobi converges quadratically: a sweep that starts below
// 3e-4 ends at rounding level), bounded by MAX_SWEEPS.
constexpr float JAC_TOL2 = 1e-15f;   // skip a rotation below cos = 3.2e-8
constexpr float JAC_CONV2 = 1e-7f;   // "converged" sweep: max cos < 3.2e-4
constexpr int JAC_MAX_SWEEPS = 12;
constexpr int JAC_DEFI_FROM = 6;     // sweeps after which a rank-deficient tile stops counting as "not converged"
// Singular VALUES alone converge one order ahead of the vectors: after a sweep that saw
// max cos c the columns are orthogonal to ~c^2 and |b_i| = s_i (1 + O(c^4)), so the
// sigma-only kernels (extract, detect, K2) may stop at c < 3.2e-2 - their sigma error
// stays at the float32 rounding floor (tools/conv_study.cpp: 8.6e-7 s_1 for every
// threshold from 1e-7 to 1e-3) with 4.1 instead of 5.0 sweeps per wave.  The embed keeps
// JAC_CONV2: its reconstruction needs the VECTORS (B orthogonal to 1e-7).
constexpr float JAC_CONV2_SIGMA = 1e-3f;
// From the 5th sweep on the embed's sweeps test every pair before rotating it: by then 93 % of the
// pair visits are below cos^2 = 1e-12 in all 64 tiles of a wave (tools/skip_study.cpp), i.e. the 5th
// sweep is almost pure verification, and a skipped pair costs its dot product only.  1e-12 is
// cos = 1e-6, an order above the float32 floor the full sweeps end at (1.2e-7): measured against the
// round-1 kernel the stego differs by 1 LSB on 4.5e-6 of the pixels (profiles/r02_embed_variants.md).
constexpr float JAC_SKIP2 = 1e-12f;

template <bool WITH_V>
WM_HD void jacobi_rot(float (&a)[8][8], float (&v)[8][8], float (&n2)[8],
                      const int p, const int q, bool& notconv) {
  float g = a[0][p] * a[0][q];
#pragma unroll
  for (int r = 1; r < 8; ++r) g = ffma(a[r][p], a[r][q], g);
  const float al = n2[p], be = n2[q];
  const float ab = al * be, gg = g * g;
  const bool big = gg > JAC_TOL2 * ab;
  notconv = notconv || (gg > JAC_CONV2 * ab);
  const float tau = be - al, g2 = g + g;
  const float h = fsqrt(ffma(tau, tau, g2 * g2));
  float t = g2 * frcp(fabsf(tau) + h);     // 0/0 only when g == 0 -> !big
  t = (tau < 0.0f) ? -t : t;
  t = big ? t : 0.0f;
  const float c = frsq(ffma(t, t, 1.0f)), s = c * t;
  const float tg = t * g;
  const float aln = al - tg, ben = be + tg;
  const bool sw = tau > 0.0f;              // |a_q| > |a_p|: rotate a further 90 deg
  const float C = sw ? s : c, S = sw ? -c : s;
  n2[p] = sw ? ben : aln;
  n2[q] = sw ? aln : ben;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float x = a[r][p], y = a[r][q];
    a[r][p] = ffma(C, x, -S * y);
    a[r][q] = ffma(S, x, C * y);
  }
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float x = v[r][p], y = v[r][q];
      v[r][p] = ffma(C, x, -S * y);
      v[r][q] = ffma(S, x, C * y);
    }
  }
}

WM_HD void col_norms2(const float (&a)[8][8], float (&n2)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float s = a[0][c] * a[0][c];
#pragma unroll
    for (int r = 1; r < 8; ++r) s = ffma(a[r][c], a[r][c], s);
    n2[c] = s;
  }
}

template <bool WITH_V>
WM_HD int jacobi_svd8(float (&a)[8][8], float (&v)[8][8], float (&n2)[8], float (&vn2)[8]) {
  if (WITH_V) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) v[r][c] = (r == c) ? 1.0f : 0.0f;
  }
  int sweep = 0;
  bool more = true;
  while (more && sweep < JAC_MAX_SWEEPS) {
    col_norms2(a, n2);
    bool notconv = false;
#pragma unroll
    for (int p = 0; p < 7; ++p)
#pragma unroll
      for (int q = p + 1; q < 8; ++q) jacobi_rot<WITH_V>(a, v, n2, p, q, notconv);
    ++sweep;
    more = wave_any(notconv);
  }
  col_norms2(a, n2);
  if (WITH_V) col_norms2(v, vn2);
  else {
#pragma unroll
    for (int c = 0; c < 8; ++c) vn2[c] = 1.0f;
  }
  return more ? -sweep : sweep;   // negative: sweep bound hit before convergence
}

// ============================================================================
// Packed (v_pk_*_f32) V-free formulation - the production embed/sigma path.
//
// Two identities remove ~45 % of the arithmetic of the literal chain
// dct2 -> svd -> U diag(S') V^T -> idct2 without changing its result beyond
// float32 rounding (parity is checked against the literal oracle):
//  (1) the DCT is orthonormal:  svd(D X D^T) = (D Ux) S (D Vx)^T, so the
//      singular values of the DCT plane ARE those of the pixel tile and
//      idct2(Uc S' Vc^T) = Ux S' Vx^T: forward and inverse DCT cancel.
//  (2) with S' = S + w (w_i = alpha*sw_i for i < K):
//      Ux S' Vx^T = X + Ux diag(w_i / s_i) Ux^T X,  because Vx^T = S^-1 Ux^T X.
//      One-sided Jacobi delivers B = X V = Ux S directly, so V is never formed.
// Rounding noise of (2) is (w_i/s_i) * eps * |X|; tiles whose smallest singular
// value is below SIGMA_RATIO_MIN * s_1 (flat / rank-deficient tiles) take the
// literal DCT-domain path (embed_tile) instead.
//
// Register layout: rows are packed in pairs, a[rp][c] = (x[2rp][c], x[2rp+1][c]),
// so a column rotation is the same packed op on 4 registers per column.
// ============================================================================
typedef float v2f __attribute__((vector_size(8)));

WM_HD v2f splat2(float s) { v2f r = {s, s}; return r; }
WM_HD v2f fma2(v2f a, v2f b, v2f c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_elementwise_fma(a, b, c);
#else
  v2f r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
  return r;
#endif
}

// dct8 / idct8 on packed row pairs (both rows of a pair at once): the row pass of the 2-D transform in the layout
// the iteration works in.  Same formulas as the scalar forms above.
WM_HD void dct8_pk(v2f& x0, v2f& x1, v2f& x2, v2f& x3, v2f& x4, v2f& x5, v2f& x6, v2f& x7) {
  const v2f s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
  const v2f d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
  const v2f e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
  x0 = splat2(C4) * (e0 + e1);
  x4 = splat2(C4) * (e0 - e1);
  x2 = fma2(splat2(C2), e2, splat2(C6) * e3);
  x6 = fma2(splat2(C6), e2, splat2(-C2) * e3);
  x1 = fma2(splat2(C1), d0, fma2(splat2(C3), d1, fma2(splat2(C5), d2, splat2(C7) * d3)));
  x3 = fma2(splat2(C3), d0, fma2(splat2(-C7), d1, fma2(splat2(-C1), d2, splat2(-C5) * d3)));
  x5 = fma2(splat2(C5), d0, fma2(splat2(-C1), d1, fma2(splat2(C7), d2, splat2(C3) * d3)));
  x7 = fma2(splat2(C7), d0, fma2(splat2(-C5), d1, fma2(splat2(C3), d2, splat2(-C1) * d3)));
}
WM_HD void idct8_pk(v2f& x0, v2f& x1, v2f& x2, v2f& x3, v2f& x4, v2f& x5, v2f& x6, v2f& x7) {
  const v2f p = splat2(C4) * (x0 + x4), q = splat2(C4) * (x0 - x4);
  const v2f r = fma2(splat2(C2), x2, splat2(C6) * x6), t = fma2(splat2(C6), x2, splat2(-C2) * x6);
  const v2f e0 = p + r, e3 = p - r, e1 = q + t, e2 = q - t;
  const v2f o0 = fma2(splat2(C1), x1, fma2(splat2(C3), x3, fma2(splat2(C5), x5, splat2(C7) * x7)));
  const v2f o1 = fma2(splat2(C3), x1, fma2(splat2(-C7), x3, fma2(splat2(-C1), x5, splat2(-C5) * x7)));
  const v2f o2 = fma2(splat2(C5), x1, fma2(splat2(-C1), x3, fma2(splat2(C7), x5, splat2(C3) * x7)));
  const v2f o3 = fma2(splat2(C7), x1, fma2(splat2(-C5), x3, fma2(splat2(C3), x5, splat2(-C1) * x7)));
  x0 = e0 + o0; x7 = e0 - o0;
  x1 = e1 + o1; x6 = e1 - o1;
  x2 = e2 + o2; x5 = e2 - o2;
  x3 = e3 + o3; x4 = e3 - o3;
}
// dct2 of a[row][col] into the packed form b[rp][c] = (C[2 rp][c], C[2 rp + 1][c]): column pass on the scalars, row pass
// on the pairs; and back.  (Packing the result of dct8x8 instead makes the compiler re-pair through scratch memory.)
WM_HD void dct8x8_to_pk(const float (&a)[8][8], v2f (&b)[4][8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {               // column pass, results straight into the pairs
    float x0 = a[0][c], x1 = a[1][c], x2 = a[2][c], x3 = a[3][c], x4 = a[4][c], x5 = a[5][c], x6 = a[6][c], x7 = a[7][c];
    dct8(x0, x1, x2, x3, x4, x5, x6, x7);
    const v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    b[0][c] = p0; b[1][c] = p1; b[2][c] = p2; b[3][c] = p3;
  }
#pragma unroll
  for (int rp = 0; rp < 4; ++rp)
    dct8_pk(b[rp][0], b[rp][1], b[rp][2], b[rp][3], b[rp][4], b[rp][5], b[rp][6], b[rp][7]);
}
WM_HD void idct8x8_from_pk(v2f (&b)[4][8], float (&a)[8][8]) {
#pragma unroll
  for (int rp = 0; rp < 4; ++rp)
    idct8_pk(b[rp][0], b[rp][1], b[rp][2], b[rp][3], b[rp][4], b[rp][5], b[rp][6], b[rp][7]);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float x0 = b[0][c][0], x1 = b[0][c][1], x2 = b[1][c][0], x3 = b[1][c][1];
    float x4 = b[2][c][0], x5 = b[2][c][1], x6 = b[3][c][0], x7 = b[3][c][1];
    idct8(x0, x1, x2, x3, x4, x5, x6, x7);
    a[0][c] = x0; a[1][c] = x1; a[2][c] = x2; a[3][c] = x3; a[4][c] = x4; a[5][c] = x5; a[6][c] = x6; a[7][c] = x7;
  }
}

// acc + (ab[HALF], ab[HALF]) * x  and  (ab[HALF], ab[HALF]) * x : the broadcast is the packed
// instruction's op_sel, written out so that the compiler cannot materialise (and then hoist and
// spill) the 64 broadcast pairs of B the embed epilogue reads.
template <int HALF>
WM_HD v2f fma2_bcast(const v2f ab, const v2f x, v2f acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(ab), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(ab), "v"(x));
  return acc;
#else
  return fma2(splat2(ab[HALF]), x, acc);
#endif
}
template <int HALF>
WM_HD v2f mul2_bcast(const v2f ab, const v2f x) {
#if defined(__HIP_DEVICE_COMPILE__)
  v2f r;
  if (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(ab), "v"(x));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(ab), "v"(x));
  return r;
#else
  return splat2(ab[HALF]) * x;
#endif
}

constexpr float SIGMA_RATIO_MIN2 = 1e-10f;   // (s_8 / s_1)^2 below this -> literal path

// one Jacobi rotation of columns p,q (no V).  c0 = cos, s0 = sin*sign(g) from
// two v_rsq_f32:  cos^2 = (1 + |tau|/h)/2,  sin = g / (h cos),  h^2 = tau^2+4g^2.
// CHECK: 0 = no convergence test, 1 = JAC_CONV2 (vectors needed), 2 = JAC_CONV2_SIGMA.
template <int CHECK, bool SKIP = false>
WM_HD void jacobi_rot_pk(v2f (&a)[4][8], float (&n2)[8], const int p, const int q, bool& notconv) {
  v2f gv = a[0][p] * a[0][q];
#pragma unroll
  for (int rp = 1; rp < 4; ++rp) gv = fma2(a[rp][p], a[rp][q], gv);
  const float g = gv[0] + gv[1];
  const float al = n2[p], be = n2[q];
  if (CHECK) notconv = notconv || (g * g > (CHECK == 2 ? JAC_CONV2_SIGMA : JAC_CONV2) * (al * be));
  // late sweeps: a pair that is below JAC_SKIP2 (cos^2) in every tile of the wave is left alone
  // (wave-uniform branch; the dot product and the test are all it costs)
  if (SKIP && !wave_any(g * g > JAC_SKIP2 * (al * be))) return;
  const float tau = be - al;
  const float ta = fabsf(tau) + 1e-18f;          // keeps 0/0 out: g == 0 -> cos = 1 exactly
  const float g2 = g + g;
  const float ih = frsq(ffma(g2, g2, ta * ta));  // 1/h
  const float x = ffma(0.5f * ta, ih, 0.5f);     // cos^2 in [0.5, 1]
  const float rx = frsq(x);
  const float c0 = x * rx;
  const float s0 = (g * ih) * rx;
  const bool sw = tau > 0.0f;                    // de Rijk: larger column ends in p
  const float C = sw ? s0 : c0, Sn = sw ? c0 : s0;
  const float w = fabsf((s0 * rx) * g);          // |t * g|: the larger norm grows by it
  n2[p] = fmaxf(al, be) + w;                     // (cancellation-free, unlike (al+be+-h)/2)
  n2[q] = fminf(al, be) - w;
  const v2f Cv = splat2(C), Sv = splat2(Sn);
#pragma unroll
  for (int rp = 0; rp < 4; ++rp) {
    const v2f X = a[rp][p], Y = a[rp][q];
    a[rp][p] = fma2(Cv, X, Sv * Y);
    a[rp][q] = fma2(Cv, Y, -(Sv * X));
  }
}

WM_HD void col_norms2_pk(const v2f (&a)[4][8], float (&n2)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v2f s = a[0][c] * a[0][c];
#pragma unroll
    for (int rp = 1; rp < 4; ++rp) s = fma2(a[rp][c], a[rp][c], s);
    n2[c] = s[0] + s[1];
  }
}

// B = X V with orthogonal columns sorted by norm; n2 = |b_i|^2.  Returns the
// sweep count (negative: bound hit).
template <int CHECK, bool SKIP = false>
WM_HD void jacobi_sweep_pk(v2f (&a)[4][8], float (&n2)[8], bool& notconv) {
#pragma unroll
  for (int p = 0; p < 7; ++p)
#pragma unroll
    for (int q = p + 1; q < 8; ++q) jacobi_rot_pk<CHECK, SKIP>(a, n2, p, q, notconv);
}

template <bool SIGMA_ONLY = false>
WM_HD int jacobi_cols_pk(v2f (&a)[4][8], float (&n2)[8]) {
  // Sweeps 1 and 2 carry no convergence test (a sweep can only be the last one if
  // it tested every pair, so these two are never last; on image tiles the earliest
  // last sweep is the 4th).  Column norms are recomputed before odd sweeps and
  // tracked through the even ones.
  bool notconv = false;
  col_norms2_pk(a, n2);
  jacobi_sweep_pk<0>(a, n2, notconv);
  jacobi_sweep_pk<0>(a, n2, notconv);
  int sweep = 2;
#if !defined(WM_EXP_R1_SWEEPS)
  // The embed's third sweep carries no test either: no image or noise tile is done after three
  // sweeps at cos^2 <= 1e-7 (tools/skip_study.cpp: 0 of 19 200), so the earliest last sweep is the 4th.
  if (!SIGMA_ONLY) {
    col_norms2_pk(a, n2);
    jacobi_sweep_pk<0>(a, n2, notconv);
    sweep = 3;
  }
#endif
  bool more = true;
  while (more && sweep < JAC_MAX_SWEEPS) {
    if ((sweep & 1) == 0) col_norms2_pk(a, n2);
    notconv = false;
#if !defined(WM_EXP_R1_SWEEPS)
    if (!SIGMA_ONLY && sweep >= 4) jacobi_sweep_pk<1, true>(a, n2, notconv);
    else
#endif
    jacobi_sweep_pk<SIGMA_ONLY ? 2 : 1>(a, n2, notconv);
    ++sweep;
    // a rank-deficient tile's null columns are rounding residue whose cosines never fall (and whose tracked norms
    // cancel to garbage): after JAC_DEFI_FROM sweeps such a lane no longer keeps its wave iterating (gen_jacobi_asm.py)
    if (sweep >= JAC_DEFI_FROM && !(n2[7] > SIGMA_RATIO_MIN2 * n2[0])) notconv = false;
    more = wave_any(notconv);
  }
  col_norms2_pk(a, n2);
  return more ? -sweep : sweep;
}

// raw tile: 8 rows x (lo, hi) little-endian byte words, as loaded from memory
struct RawTile { uint32_t lo[8], hi[8]; };

WM_HD float raw_px(const RawTile& t, const int r, const int c) {
  const uint32_t w = (c < 4) ? t.lo[r] : t.hi[r];
  return (float)((w >> (8 * (c & 3))) & 0xffu);
}
WM_HD void raw_to_pk(const RawTile& t, v2f (&a)[4][8]) {
#pragma unroll
  for (int rp = 0; rp < 4; ++rp)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v2f v = {raw_px(t, 2 * rp, c), raw_px(t, 2 * rp + 1, c)};
      a[rp][c] = v;
    }
}
WM_HD void raw_to_f32(const RawTile& t, float (&a)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) a[r][c] = raw_px(t, r, c);
}

// singular values of a raw uint8 tile (pixel domain == DCT domain, identity (1))
WM_HD int sigma_tile_pk(const RawTile& t, float (&s)[8]) {
  v2f a[4][8];
  float n2[8];
  raw_to_pk(t, a);
  const int sweeps = jacobi_cols_pk<true>(a, n2);
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = fsqrt(n2[i]);
  return sweeps;
}

// ---- uint8 quantisation: np.clip(x, 0, 255).astype(np.uint8) ----------------
WM_HD uint32_t quant_u8(float x) {
  x = fminf(fmaxf(x, 0.0f), 255.0f);   // NaN -> 0 via fmaxf
  return (uint32_t)x;                  // truncation toward zero
}

// V-free embed of a raw tile.  `out` receives the quantised stego bytes, sc the
// host singular values Sc; when YW is true the unclipped float stego is written
// through `yw` (row stride `yw_stride` floats).  Returns sweeps; `deficient` is
// set when the tile must take the literal path instead.
// Phase 1 (registers: B only): B = X V from the raw tile.
WM_HD int embed_jacobi_pk(const RawTile& t, v2f (&a)[4][8], float (&n2)[8]) {
  raw_to_pk(t, a);
  return jacobi_cols_pk(a, n2);
}

// Phase 2a: singular values out, e_i = alpha_i sw_i / s_i^3, rank-deficiency flag.
WM_HD void embed_coeffs_pk(const float (&n2)[8], const float (&sw)[8], const float (&alpha_k)[8],
                           float (&e)[8], float (&sc)[8], bool& deficient) {
  deficient = !(n2[7] > SIGMA_RATIO_MIN2 * n2[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float rs = frsq(fmaxf(n2[i], 1e-30f));
    sc[i] = n2[i] * rs;
    e[i] = (alpha_k[i] * sw[i]) * (rs * rs * rs);
  }
}

// Phase 2b: four columns of Y = X + B G, G = diag(e) (B^T X), from the eight row words `w`
// (little-endian bytes = the four pixels of each row in this half), one column pair at a time
// so that only 8 packed registers of G are live next to B:  g[i] = (G[i][c], G[i][c+1]);
// b_i[r] is one half of a[r>>1][i], broadcast by the packed FMA's op_sel.  Columns of X only
// enter their own columns of Y, so the two halves of a tile are independent (the kernel loads,
// finishes and stores them one after the other to stay within 128 VGPRs).
// outw: quantised stego bytes; with YW the unclipped floats go to yw[r * yw_stride + 0..3].
template <bool YW>
WM_HD void embed_half_pk(const uint32_t (&w)[8], const v2f (&a)[4][8], const float (&e)[8],
                         uint32_t (&outw)[8], float* yw, const size_t yw_stride) {
#pragma unroll
  for (int r = 0; r < 8; ++r) outw[r] = 0u;
#pragma unroll
  for (int cp = 0; cp < 2; ++cp) {
    const int sh = 16 * cp;
    v2f g[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t ww = w[r] >> sh;
      const v2f x = {(float)(ww & 0xffu), (float)((ww >> 8) & 0xffu)};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v2f ab = a[r >> 1][i];
        if (r == 0) g[i] = mul2_bcast<0>(ab, x);
        else g[i] = (r & 1) ? fma2_bcast<1>(ab, x, g[i]) : fma2_bcast<0>(ab, x, g[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = g[i] * splat2(e[i]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t ww = w[r] >> sh;
      v2f y = {(float)(ww & 0xffu), (float)((ww >> 8) & 0xffu)};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v2f ab = a[r >> 1][i];
        y = (r & 1) ? fma2_bcast<1>(ab, g[i], y) : fma2_bcast<0>(ab, g[i], y);
      }
      if (YW) {
        float* o = yw + (size_t)r * yw_stride + 2 * cp;
        o[0] = y[0]; o[1] = y[1];
      }
      outw[r] |= (quant_u8(y[0]) | (quant_u8(y[1]) << 8)) << sh;
    }
  }
}

// Phase 2 in one piece (CPU harness / simple callers): the raw tile in, the stego tile out.
template <bool YW>
WM_HD void embed_finish_pk(const RawTile& t, const v2f (&a)[4][8], const float (&n2)[8],
                           const float (&sw)[8], const float (&alpha_k)[8], float (&sc)[8],
                           RawTile& out, float* yw, const size_t yw_stride, bool& deficient) {
  float e[8];
  embed_coeffs_pk(n2, sw, alpha_k, e, sc, deficient);
  embed_half_pk<YW>(t.lo, a, e, out.lo, yw, yw_stride);
  embed_half_pk<YW>(t.hi, a, e, out.hi, YW ? yw + 4 : nullptr, yw_stride);
}

// both phases back to back (CPU harness / simple callers)
template <bool YW>
WM_HD int embed_tile_pk(const RawTile& t, const float (&sw)[8], const float (&alpha_k)[8],
                        float (&sc)[8], RawTile& out, float* yw, const size_t yw_stride,
                        bool& deficient) {
  v2f a[4][8];
  float n2[8];
  const int sweeps = embed_jacobi_pk(t, a, n2);
  embed_finish_pk<YW>(t, a, n2, sw, alpha_k, sc, out, yw, yw_stride, deficient);
  return sweeps;
}


// ---- embed: tile (already float, pixel domain) -> stego tile (float) --------
// sw[8]: the watermark tile's singular values; alpha_k[i] = alpha for i < K,
// 0 otherwise.  sc[8] receives the host tile's singular values.
WM_HD int embed_tile(float (&a)[8][8], const float (&sw)[8], const float (&alpha_k)[8],
                     float (&sc)[8]) {
  float v[8][8], n2[8], vn2[8];
  dct8x8(a);
  const int sweeps = jacobi_svd8<true>(a, v, n2, vn2);
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float nb = fsqrt(n2[i]), nv = fsqrt(vn2[i]);
    const float sig = nb * frcp(nv);                      // sigma_i = |b_i| / |v_i|
    sc[i] = sig;
    const float sp = ffma(alpha_k[i], sw[i], sig);        // S_[:K] = Sc[:K] + alpha*Sw[:K]
    const float den = nb * nv;
    f[i] = (den > 0.0f) ? sp * frcp(den) : 0.0f;          // sigma'_i / (|b_i| |v_i|)
  }
  // Cw = sum_i (b_i f_i) v_i^T   ==  U diag(S_) V^T
  float cw[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bs[i] = a[r][i] * f[i];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float s = bs[0] * v[c][0];
#pragma unroll
      for (int i = 1; i < 8; ++i) s = ffma(bs[i], v[c][i], s);
      cw[r][c] = s;
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) a[r][c] = cw[r][c];
  idct8x8(a);
  return sweeps;
}

// ---- rank-deficient tiles: deterministic orthonormal completion ---------------
// A flat / saturated tile has singular values that are exactly 0; its singular
// vectors there are arbitrary (LAPACK returns *some* orthonormal completion) but
// the scheme still injects alpha*sw_i along them (single:175-176).  One-sided
// Jacobi resolves directions to high *relative* accuracy, so adding
// COMPLETION_DELTA x a fixed full-rank pattern (cond 3.7) to the DCT tile makes
// U and V complete orthonormal bases; the pattern is subtracted again after
// the reconstruction, so the output moves by O(eps), not O(delta).
constexpr float COMPLETION_DELTA = 6.103515625e-05f;   // 2^-14
constexpr float COMPLETION_PATTERN[8][8] = {
  {-0.371394f, -0.079717f, -0.983650f, -0.554458f, -0.658720f, -0.953876f, +0.008741f, +0.773822f},
  {+0.770895f, +0.280528f, -0.063239f, +0.001315f, +0.675874f, -0.488476f, +0.610478f, +0.151159f},
  {+0.434160f, -0.517163f, -0.154770f, +0.779245f, -0.744333f, +0.573436f, +0.871420f, -0.682901f},
  {-0.025223f, -0.797351f, +0.189486f, -0.674952f, -0.442962f, +0.883550f, -0.294998f, +0.418505f},
  {-0.560398f, +0.910414f, +0.028949f, -0.670469f, -0.276553f, +0.424304f, +0.871431f, -0.253670f},
  {-0.405084f, +0.378566f, +0.483330f, +0.296345f, -0.562919f, -0.634010f, -0.141621f, +0.624869f},
  {+0.385623f, -0.252198f, +0.255703f, +0.219000f, -0.437414f, -0.709251f, -0.041911f, -0.468894f},
  {+0.789973f, +0.864080f, -0.920759f, -0.063243f, +0.524377f, -0.040678f, +0.003126f, -0.226323f}};

#include "wm_completion_tables.inc"     // CONST_TILE_S / CONST_TILE_M (tools/gen_completion_tables.py)

// ---- constant tiles (letterbox bars, flat backgrounds): the literal chain in closed form -------------
// Every pixel = v: dct2 is 8 v E00, and to first order in delta / (8 v) the SVD of 8 v E00 + delta P is the
// pattern's own (v = 0) or e0 followed by that of P[1:, 1:] (v > 0) - fixed vectors, so the chain
// "+ delta P -> svd -> U diag(S + alpha Sw) V^T -> - delta P -> idct2" is  v + alpha sum_i sw_i M_i  with the tabulated
// pixel-domain matrices M_i, and Sc = (8 v, 0, ..) + delta * (tabulated values).  Same completion as
// embed_tile_completed (which such a tile would otherwise go through, at ~40 k instructions per wave) to
// ~1e-2 grey levels; a constant tile takes THIS path whatever shares its wave, so results stay reproducible.
WM_HD bool raw_is_constant(const RawTile& t) {
  const uint32_t w = (t.lo[0] & 0xffu) * 0x01010101u;
  uint32_t diff = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) diff |= (t.lo[r] ^ w) | (t.hi[r] ^ w);
  return diff == 0;
}
WM_HD bool tile_is_constant(const float (&a)[8][8]) {
  bool same = true;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) same = same && (a[r][c] == a[0][0]);
  return same;
}
// out <- Yw of the constant tile of value v; sc <- its (completed) singular values
WM_HD void embed_tile_constant(const float v, const float (&sw)[8], const float (&alpha_k)[8], float (&sc)[8],
                               float (&out)[8][8]) {
  const bool black = v == 0.0f;
  float w0[8], w1[8];                       // weights of the v == 0 / v > 0 tables: one of the two sets is zero
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float w = alpha_k[i] * sw[i];
    w0[i] = black ? w : 0.0f;
    w1[i] = black ? 0.0f : w;
    sc[i] = black ? CONST_TILE_S[0][i] : CONST_TILE_S[1][i];
  }
  sc[0] = ffma(8.0f, v, sc[0]);
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = v;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = ffma(w0[i], CONST_TILE_M[0][i][r][c], ffma(w1[i], CONST_TILE_M[1][i][r][c], acc));
      out[r][c] = acc;
    }
}

// the same with the table known at compile time (a wave whose constant tiles are all black, or none of them):
// bit-identical to embed_tile_constant, whose other table only ever adds 0 * M
template <int TABLE>
WM_HD void embed_tile_constant_t(const float v, const float (&sw)[8], const float (&alpha_k)[8], float (&sc)[8],
                                 float (&out)[8][8]) {
  float w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    w[i] = alpha_k[i] * sw[i];
    sc[i] = CONST_TILE_S[TABLE][i];
  }
  sc[0] = ffma(8.0f, v, sc[0]);
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = v;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = ffma(w[i], CONST_TILE_M[TABLE][i][r][c], acc);
      out[r][c] = acc;
    }
}

// ---- rank-1 tiles: the literal chain in closed form ------------------------------------------------------------
// X = a b^T with a, b >= 0: the edges of flat rectangles (rows or columns all equal), 1-pixel rules and their crossings
// on a flat background, one-axis gradients - 34 % + 4 % of the tiles of the synthetic screen content against 0.5 % of
// rank 2.  sigma_1 = |a| |b|, u_1 = a / |a|, v_1 = b / |b|; the other seven singular values are 0 and their vectors
// arbitrary (LAPACK returns SOME orthonormal completion; any one gives the reference's svd(Yw) = Sc + alpha Sw,
// single:174-176).  The completion taken here, in the pixel domain (the DCT is orthonormal and cancels):
//   U = H_u diag(-1, 1, ..),  V = H_v diag(-1, 1, ..),   H_u = I - beta_u p p^T with p = u_1 + e_0, beta_u = 1 / (1 + u_1[0])
//   (a Householder reflection with H_u e_0 = -u_1; pixels are >= 0, so u_1[0] >= 0 and p never cancels), H_v likewise from v_1:
//   Yw = U diag(Sc + w) V^T = X + H_u diag(w) H_v,      w = alpha Sw[:K]
//      = X + diag(w) - beta_u p (p o w)^T - beta_v (w o q) q^T + beta_u beta_v (p^T diag(w) q) p q^T
// - four FMAs per pixel instead of a Jacobi with V (~40 000 instructions per wave).  a and b are read off the row and
// column sums (R = a sum(b), C = b sum(a)): u_1 = R / |R|, v_1 = C / |C|, sigma_1 = |R| |C| / sum(X).
// Rank 1 is decided EXACTLY on the integer pixels: X_ij * S == R_i * C_j for every i, j (S = sum(X) > 0).
WM_HD bool raw_rank1_pretest(const RawTile& t) {      // three 2x2 minors: textured tiles almost never pass, rank-1 tiles always
  const uint32_t x00 = t.lo[0] & 0xffu, x04 = t.hi[0] & 0xffu, x07 = t.hi[0] >> 24;
  const uint32_t x33 = t.lo[3] >> 24, x34 = t.hi[3] & 0xffu;
  const uint32_t x40 = t.lo[4] & 0xffu, x43 = t.lo[4] >> 24, x44 = t.hi[4] & 0xffu;
  const uint32_t x70 = t.lo[7] & 0xffu, x77 = t.hi[7] >> 24;
  return x00 * x77 == x07 * x70 && x33 * x44 == x34 * x43 && x00 * x44 == x04 * x40;
}
WM_HD bool raw_is_rank1(const RawTile& t) {
  uint32_t R[8], Cs[8], S = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) Cs[j] = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t b = ((j < 4 ? t.lo[r] : t.hi[r]) >> (8 * (j & 3))) & 0xffu;
      s += b; Cs[j] += b;
    }
    R[r] = s; S += s;
  }
  uint32_t bad = (S == 0) ? 1u : 0u;                  // the zero tile is a constant tile, not a rank-1 one
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t b = ((j < 4 ? t.lo[r] : t.hi[r]) >> (8 * (j & 3))) & 0xffu;
      bad |= (b * S) ^ (R[r] * Cs[j]);                // <= 255 * 16 320 and 2 040^2: no overflow
    }
  return bad == 0;
}
// x: the tile's pixels (rank 1, not zero); out <- Yw, sc <- singular values.  out may alias x.
WM_HD void embed_tile_rank1(const float (&x)[8][8], const float (&sw)[8], const float (&alpha_k)[8], float (&sc)[8],
                            float (&out)[8][8]) {
  float R[8], Cs[8], S = 0.0f;                        // sums of integers <= 16 320: exact in float
#pragma unroll
  for (int j = 0; j < 8; ++j) Cs[j] = 0.0f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += x[r][j]; Cs[j] += x[r][j]; }
    R[r] = s; S += s;
  }
  float nr2 = 0.0f, nc2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { nr2 = ffma(R[i], R[i], nr2); nc2 = ffma(Cs[i], Cs[i], nc2); }
  const float rr = frsq(nr2), rc = frsq(nc2);
  float p[8], q[8], w[8], pw[8], wq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = R[i] * rr; q[i] = Cs[i] * rc; w[i] = alpha_k[i] * sw[i]; sc[i] = 0.0f; }
  p[0] += 1.0f; q[0] += 1.0f;
  sc[0] = (nr2 * rr) * (nc2 * rc) * frcp(S);          // |R| |C| / S
  const float bu = frcp(p[0]), bv = frcp(q[0]);
  float gamma = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { pw[i] = p[i] * w[i]; wq[i] = w[i] * q[i]; gamma = ffma(pw[i], q[i], gamma); }
  const float bg = bu * bv * gamma;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float c_p = -bu * p[i], c_q = -bv * wq[i], c_pq = bg * p[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = x[i][j] + ((i == j) ? w[i] : 0.0f);
      v = ffma(c_p, pw[j], v);
      v = ffma(c_q, q[j], v);
      v = ffma(c_pq, q[j], v);
      out[i][j] = v;
    }
  }
}

// ---- tiles with exactly ONE singular value out of the V-free form's reach (s_7 > 1e-5 s_1 >= s_8) --------------
// On noise and camera content this is what the flagged tiles are (~50 per 4K frame): full rank but with s_8 so small
// that  e_8 = w_8 / s_8^3  and  b_8^T X  cancel catastrophically - or rank 7 exactly.  They do not need the literal chain
// (Jacobi with V, ~40 k instructions per wave): the seven good vectors of either side come from the fast path's own
// B = X V (u_i = b_i / |b_i|, v_i = X^T b_i / s_i^2), and the eighth of either side is DETERMINED by them - the orthogonal
// complement of seven vectors in R^8 is a line:  P = I - sum_i q_i q_i^T  is  q_8 q_8^T, so its column with the largest
// diagonal entry (>= 1/8), orthogonalised once more and normalised, is q_8 up to sign (better than b_8 / |b_8|, whose
// direction is only good to eps s_1 / s_8).  The joint sign of (u_8, v_8) is that of u_8^T X v_8 = +-s_8, evaluated in
// FLOAT64: the vectors' float32 errors enter it only in second order (each is orthogonal to its side's other seven), so
// the sign is right down to s_8 ~ 1e-12 s_1 - where the reference's float64 LAPACK stops resolving it too.  With
// s_8 == 0 exactly the reference's pair is arbitrary and so is this one.   Yw = X + sum_i alpha sw_i u_i v_i^T, Sc = |b_i|.
// bb[r][i] = component r of b_i (the layout of the fast kernel's packed a[r >> 1][i][r & 1]).
WM_HD bool n2_one_small(const float (&n2)[8]) { return n2[6] > SIGMA_RATIO_MIN2 * n2[0]; }
// r <- unit vector orthogonal to q_0 .. q_6, q_i[a] = Q(i, a) * scale[i]
template <typename QF>
WM_HD void complement_of_seven(const QF& Q, float (&r)[8]) {
  float diag[8];                                   // of P = I - sum q_i q_i^T: only the diagonal and ONE column are formed
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float acc = 1.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc = ffma(-Q(i, a), Q(i, a), acc);
    diag[a] = acc;
  }
  int best = 0;
  float dbest = diag[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) { const bool gt = diag[k] > dbest; best = gt ? k : best; dbest = gt ? diag[k] : dbest; }
  float qb[7];                                     // q_i[best]
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    float v = Q(i, 0);
#pragma unroll
    for (int k = 1; k < 8; ++k) v = (best == k) ? Q(i, k) : v;
    qb[i] = v;
  }
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float acc = (best == a) ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc = ffma(-Q(i, a), qb[i], acc);
    r[a] = acc;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {                    // once more against the seven: the column carries their rounding
    float d = 0.0f;
#pragma unroll
    for (int a = 0; a < 8; ++a) d = ffma(r[a], Q(i, a), d);
#pragma unroll
    for (int a = 0; a < 8; ++a) r[a] = ffma(-d, Q(i, a), r[a]);
  }
  float n = 0.0f;
#pragma unroll
  for (int a = 0; a < 8; ++a) n = ffma(r[a], r[a], n);
  const float rn = frsq(n);
#pragma unroll
  for (int a = 0; a < 8; ++a) r[a] *= rn;
}
WM_HD void embed_tile_one_small(const float (&x)[8][8], const float (&bb)[8][8], const float (&sw)[8],
                                const float (&alpha_k)[8], float (&sc)[8], float (&out)[8][8]) {
  // three 8 x 8 arrays live at a time (x, B, V): u_i = b_i / |b_i| stays folded into the weights
  float v[8][8];                                   // v[i][c]
  float rs[8];                                     // 1 / |b_i|
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float n = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) n = ffma(bb[r][i], bb[r][i], n);
    rs[i] = frsq(fmaxf(n, 1e-30f));
    sc[i] = n * rs[i];
  }
  float u7[8];
  complement_of_seven([&](const int i, const int a) { return bb[a][i] * rs[i]; }, u7);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const float r2 = rs[i] * rs[i];                // v_i = X^T b_i / s_i^2
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) acc = ffma(x[r][c], bb[r][i], acc);
      v[i][c] = acc * r2;
    }
  }
  complement_of_seven([&](const int i, const int a) { return v[i][a]; }, v[7]);
  {
    double d = 0.0;                                // sigma_8 >= 0: u_8^T X v_8 must not be negative (float64: see above)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) t += (double)x[r][c] * (double)v[7][c];
      d += (double)u7[r] * t;
    }
    const float sgn = (d < 0.0) ? -1.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[7][c] *= sgn;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = ffma(alpha_k[7] * sw[7] * u7[r], v[7][c], x[r][c]);
#pragma unroll
      for (int i = 0; i < 7; ++i) acc = ffma(alpha_k[i] * sw[i] * rs[i] * bb[r][i], v[i][c], acc);
      out[r][c] = acc;
    }
}

// ---- any rank from 2 to 6: the same construction with more than one missing pair ------------------------------
// Ranks i with s_i > 1e-5 s_1 take u_i, v_i from B as above; every other rank gets a pair (u_j, v_j) from the orthogonal
// complements of what is there so far, one vector at a time (largest diagonal entry of the current projector, orthogonalised
// once more, normalised).  More than one missing direction per side: the reference's completion is arbitrary (LAPACK's
// choice of a basis of the null spaces) and so is this one; what holds whatever the choice - svd(Yw) = Sc + alpha Sw,
// Sc = the tile's singular values, the injected energy - is what the tests check.  u[i][r], v[i][c] hold zero vectors for
// the ranks not filled yet, so every sum runs over all eight slots.
WM_HD void complete_next(const float (&q)[8][8], float (&r)[8]) {
  float diag[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    float acc = 1.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = ffma(-q[i][a], q[i][a], acc);
    diag[a] = acc;
  }
  int best = 0;
  float dbest = diag[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) { const bool gt = diag[k] > dbest; best = gt ? k : best; dbest = gt ? diag[k] : dbest; }
#pragma unroll
  for (int a = 0; a < 8; ++a) r[a] = (best == a) ? 1.0f : 0.0f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float d = 0.0f;
#pragma unroll
      for (int a = 0; a < 8; ++a) d = ffma(r[a], q[i][a], d);
#pragma unroll
      for (int a = 0; a < 8; ++a) r[a] = ffma(-d, q[i][a], r[a]);
    }
  float n = 0.0f;
#pragma unroll
  for (int a = 0; a < 8; ++a) n = ffma(r[a], r[a], n);
  const float rn = frsq(n);
#pragma unroll
  for (int a = 0; a < 8; ++a) r[a] *= rn;
}
WM_HD void embed_tile_from_b(const float (&x)[8][8], const float (&bb)[8][8], const float (&sw)[8],
                             const float (&alpha_k)[8], float (&sc)[8], float (&out)[8][8]) {
  float u[8][8], v[8][8];
  bool good[8];
  float n0 = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float n = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) n = ffma(bb[r][i], bb[r][i], n);
    if (i == 0) n0 = n;
    const float rs = frsq(fmaxf(n, 1e-30f));
    sc[i] = n * rs;
    good[i] = n > SIGMA_RATIO_MIN2 * n0 && n > 0.0f;
    const float g = good[i] ? rs : 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) u[i][r] = bb[r][i] * g;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) acc = ffma(x[r][c], u[i][r], acc);
      v[i][c] = acc * g;                           // X^T u_i / s_i  (0 for the ranks to be completed)
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (!good[j]) {
      float ru[8], rv[8];
      complete_next(u, ru);
      complete_next(v, rv);
#pragma unroll
      for (int a = 0; a < 8; ++a) { u[j][a] = ru[a]; v[j][a] = rv[a]; }
    }
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float acc = x[r][c];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = ffma(alpha_k[i] * sw[i] * u[i][r], v[i][c], acc);
      out[r][c] = acc;
    }
}

WM_HD void add_completion(float (&a)[8][8], const float scale) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) a[r][c] = ffma(scale, COMPLETION_PATTERN[r][c], a[r][c]);
}

// the same on the packed form (rows 2 rp, 2 rp + 1 of column c in one v2f): what the literal chain uses, so that the
// pattern is added in the layout the iteration works in (added to a[r][c] the compiler pairs neighbouring COLUMNS
// and re-pairs them through scratch memory)
WM_HD void add_completion_pk(v2f (&b)[4][8], const float scale) {
#pragma unroll
  for (int rp = 0; rp < 4; ++rp)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const v2f pt = {COMPLETION_PATTERN[2 * rp][c], COMPLETION_PATTERN[2 * rp + 1][c]};
      b[rp][c] = fma2(splat2(scale), pt, b[rp][c]);
    }
}

// ---- packed one-sided Jacobi WITH V: A and V stacked as one 16-row matrix ----
// The same rotation (jacobi_rot_pk's two-rsq angle) is applied to the 4 row pairs
// of A and the 4 row pairs of V; dot products and norms come from the A half only.
template <int CHECK>
WM_HD void jacobi_rot_pk_v(v2f (&a)[4][8], v2f (&v)[4][8], float (&n2)[8], const int p, const int q,
                           bool& notconv) {
  v2f gv = a[0][p] * a[0][q];
#pragma unroll
  for (int rp = 1; rp < 4; ++rp) gv = fma2(a[rp][p], a[rp][q], gv);
  const float g = gv[0] + gv[1];
  const float al = n2[p], be = n2[q];
  if (CHECK) notconv = notconv || (g * g > JAC_CONV2 * (al * be));
  const float tau = be - al;
  const float ta = fabsf(tau) + 1e-18f;
  const float g2 = g + g;
  const float ih = frsq(ffma(g2, g2, ta * ta));
  const float x = ffma(0.5f * ta, ih, 0.5f);
  const float rx = frsq(x);
  const float c0 = x * rx;
  const float s0 = (g * ih) * rx;
  const bool sw = tau > 0.0f;
  const float C = sw ? s0 : c0, Sn = sw ? c0 : s0;
  const float w = fabsf((s0 * rx) * g);
  n2[p] = fmaxf(al, be) + w;
  n2[q] = fminf(al, be) - w;
  const v2f Cv = splat2(C), Sv = splat2(Sn);
#pragma unroll
  for (int rp = 0; rp < 4; ++rp) {
    const v2f X = a[rp][p], Y = a[rp][q];
    a[rp][p] = fma2(Cv, X, Sv * Y);
    a[rp][q] = fma2(Cv, Y, -(Sv * X));
  }
#pragma unroll
  for (int rp = 0; rp < 4; ++rp) {
    const v2f X = v[rp][p], Y = v[rp][q];
    v[rp][p] = fma2(Cv, X, Sv * Y);
    v[rp][q] = fma2(Cv, Y, -(Sv * X));
  }
}

// B = A V (columns orthogonal, sorted by norm), V accumulated from the identity.
// n2 = |b_i|^2, vn2 = |v_i|^2 (1 up to the drift of v_rsq_f32, carried so that it
// cancels in sigma_i = |b_i| / |v_i|).  Column norms are recomputed before every
// sweep: this is the path for tiles whose trailing columns are ~1e-8 of the leading one.
#if defined(__HIPCC__) && !defined(WM_NO_ASM_JACOBI) && !defined(WM_NO_ASM_JACOBI_V)
#include "wm_jacobi_v_gfx950.inc"      // generated gfx950 stream of the iteration below (tools/gen_jacobi_asm.py)
#endif
WM_HD int jacobi_cols_pk_v(v2f (&a)[4][8], v2f (&v)[4][8], float (&n2)[8], float (&vn2)[8]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WM_NO_ASM_JACOBI) && !defined(WM_NO_ASM_JACOBI_V)
  return jacobi_cols_v_gfx950(a, v, n2, vn2, JAC_CONV2);
#else
#pragma unroll
  for (int rp = 0; rp < 4; ++rp)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v2f e = {(2 * rp == c) ? 1.0f : 0.0f, (2 * rp + 1 == c) ? 1.0f : 0.0f};
      v[rp][c] = e;
    }
  // A tile that has converged (one of its own sweeps saw nothing to rotate) is frozen: its lane
  // sits out (exec mask) the extra sweeps its wave neighbours need.  The fallback kernel groups
  // tiles in the (atomic, run-to-run varying) order of its work list; with the freeze a tile's
  // result does not depend on which tiles share its wave.
  int sweep = 0;
  bool more = true, active = true;
  while (more && sweep < JAC_MAX_SWEEPS) {
    if (active) {
      col_norms2_pk(a, n2);
      bool notconv = false;
#pragma unroll
      for (int p = 0; p < 7; ++p)
#pragma unroll
        for (int q = p + 1; q < 8; ++q) jacobi_rot_pk_v<1>(a, v, n2, p, q, notconv);
      active = notconv;
    }
    ++sweep;
    more = wave_any(active);
  }
  col_norms2_pk(a, n2);
  col_norms2_pk(v, vn2);
  return more ? -sweep : sweep;
#endif
}

// literal chain on a (possibly) rank-deficient tile: dct2 -> (+delta P) -> svd
// with V -> U diag(S + alpha Sw) V^T -> (-delta P) -> idct2
WM_HD int embed_tile_completed(float (&a)[8][8], const float (&sw)[8], const float (&alpha_k)[8],
                               float (&sc)[8]) {
  v2f b[4][8], v[4][8];
  float n2[8], vn2[8];
  dct8x8_to_pk(a, b);
  add_completion_pk(b, COMPLETION_DELTA);
  const int sweeps = jacobi_cols_pk_v(b, v, n2, vn2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float nb = fsqrt(n2[i]), nv = fsqrt(vn2[i]);
    const float sig = nb * frcp(nv);                      // sigma_i = |b_i| / |v_i|
    sc[i] = sig;
    const float sp = ffma(alpha_k[i], sw[i], sig);        // S_[:K] = Sc[:K] + alpha*Sw[:K]
    const float den = nb * nv;
    const float f = (den > 0.0f) ? sp * frcp(den) : 0.0f; // sigma'_i / (|b_i| |v_i|)
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) b[rp][i] = b[rp][i] * splat2(f);
  }
  // Cw = sum_i (b_i f_i) v_i^T   ==  U diag(S_) V^T ; row pair rp, column c
  v2f cw[4][8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
      v2f acc = b[rp][0] * splat2(v[c >> 1][0][c & 1]);
#pragma unroll
      for (int i = 1; i < 8; ++i) acc = fma2(b[rp][i], splat2(v[c >> 1][i][c & 1]), acc);
      const v2f pt = {COMPLETION_PATTERN[2 * rp][c], COMPLETION_PATTERN[2 * rp + 1][c]};
      cw[rp][c] = fma2(splat2(-COMPLETION_DELTA), pt, acc);
    }
  idct8x8_from_pk(cw, a);
  return sweeps;
}

// ---- sigma only (extract / detect): tile -> singular values -----------------
WM_HD int sigma_tile(float (&a)[8][8], float (&s)[8]) {
  float n2[8], vn2[8];
  dct8x8(a);
  const int sweeps = jacobi_svd8<false>(a, a /*unused*/, n2, vn2);
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = fsqrt(n2[i]);
  return sweeps;
}

// ---- full SVD of a float tile (watermark side): U, S, Vt --------------------
// u[r][i], vt[i][c]; columns with sigma == 0 get u_i = 0.
WM_HD int svd_tile(float (&a)[8][8], float (&s)[8], float (&vt)[8][8], const bool complete = false) {
  v2f b[4][8], v[4][8];
  float n2[8], vn2[8];
  dct8x8_to_pk(a, b);
  if (complete) add_completion_pk(b, COMPLETION_DELTA);
  const int sweeps = jacobi_cols_pk_v(b, v, n2, vn2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float rb = (n2[i] > 0.0f) ? frsq(n2[i]) : 0.0f;
    const float rv = frsq(vn2[i]);
    s[i] = fsqrt(n2[i]) * rv;
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {                      // U = B / |b_i|
      const v2f u = b[rp][i] * splat2(rb);
      a[2 * rp][i] = u[0];
      a[2 * rp + 1][i] = u[1];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) vt[i][c] = v[c >> 1][i][c & 1] * rv;   // Vt = (V / |v_i|)^T
  }
  return sweeps;
}

// ---- extract: Wm_hat = Uw diag(sw_hat) Vwt, then IDCT (a8 + a9) -------------
// s_cw/sc: stego and stored host singular values; inv_alpha = 1/max(alpha,1e-8);
// keep[i] = 1 for i < K else 0.  out <- idct(Uw diag(sw_hat) Vwt).
WM_HD void extract_tile(const float (&s_cw)[8], const float (&sc)[8], const float inv_alpha,
                        const float (&keep)[8], const float (&uw)[8][8],
                        const float (&vwt)[8][8], float (&out)[8][8]) {
  float sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sh[i] = (s_cw[i] - sc[i]) * inv_alpha * keep[i];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float us[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) us[i] = uw[r][i] * sh[i];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float s = us[0] * vwt[0][c];
#pragma unroll
      for (int i = 1; i < 8; ++i) s = ffma(us[i], vwt[i][c], s);
      out[r][c] = s;
    }
  }
  idct8x8(out);
}

// ---- the same with PIXEL-domain factors -------------------------------------
// idct2(Uw diag(s) Vwt) = (D^T Uw) diag(s) (Vwt D): with Ux = D^T Uw and Vxt = Vwt D
// prepared once per watermark (factors_to_pixel) the per-frame work is the rank-8 product
// alone - no IDCT - and it packs over column pairs.
WM_HD void factors_to_pixel(float (&u)[8][8], float (&vt)[8][8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c)     // Ux = D^T Uw: inverse transform along the row index
    idct8(u[0][c], u[1][c], u[2][c], u[3][c], u[4][c], u[5][c], u[6][c], u[7][c]);
#pragma unroll
  for (int i = 0; i < 8; ++i)     // Vxt = Vwt D: inverse transform along the column index
    idct8(vt[i][0], vt[i][1], vt[i][2], vt[i][3], vt[i][4], vt[i][5], vt[i][6], vt[i][7]);
}

WM_HD void extract_tile_px(const float (&s_cw)[8], const float (&sc)[8], const float inv_alpha,
                           const float (&keep)[8], const float (&ux)[8][8],
                           const float (&vxt)[8][8], float (&out)[8][8]) {
  float sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sh[i] = (s_cw[i] - sc[i]) * inv_alpha * keep[i];
  v2f vs[8][4];                    // diag(sh) Vxt, packed over column pairs
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
      const v2f v = {vxt[i][2 * cp], vxt[i][2 * cp + 1]};
      vs[i][cp] = v * splat2(sh[i]);
    }
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
      v2f acc = splat2(ux[r][0]) * vs[0][cp];
#pragma unroll
      for (int i = 1; i < 8; ++i) acc = fma2(splat2(ux[r][i]), vs[i][cp], acc);
      out[r][2 * cp] = acc[0];
      out[r][2 * cp + 1] = acc[1];
    }
}

}  // namespace wm
