// libwmhip.so - gfx950 kernels + C ABI (include/wmhip.h) of the tile-mode
// DCT-SVD watermark hot path.
//
// Execution model: ONE 8x8 TILE PER LANE.  A wave64 owns 64 consecutive tiles
// (row-major tile order); every load/store instruction of a wave touches
// 64 x 8 B = 512 contiguous bytes of one image row (coalesced), and the whole
// DCT -> one-sided Jacobi SVD -> perturb -> reconstruct -> IDCT -> quantise
// chain runs out of VGPRs with compile-time register indices: no LDS round
// trips, no cross-lane shuffles, no divergence (the sweep loop is
// wave-uniform).  The path is VALU-bound (~10^4 FMA-class instructions per
// tile against 3 B per pixel); see DESIGN.md for the roofline.
//
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libwmhip.so wmhip.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "wm_internal.h"
#include "wm_tile_math.h"

using namespace wmi;

namespace wmi {

thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, const char* a, const char* b) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

int grow(wm_ctx* ctx, void** buf, size_t* have, size_t bytes, const char* what) {
  if (bytes <= *have) return WM_OK;
  if (*buf) {
    WM_HIP(hipStreamSynchronize(ctx->stream));
    WM_HIP(hipFree(*buf));
    *buf = nullptr; *have = 0;
  }
  const size_t mb = (size_t)1 << 20;
  const size_t want = (bytes + mb - 1) / mb * mb;
  if (hipMalloc(buf, want) != hipSuccess) {
    (void)hipGetLastError();
    *buf = nullptr;
    return set_err(WM_ERR_NOMEM, "hipMalloc failed for %s", what);
  }
  *have = want;
  return WM_OK;
}

}  // namespace wmi

namespace {

constexpr int N_SUMS = 5;          // detect: sum a, b, ab, aa, bb
constexpr unsigned FB_SUB = 64;    // embed: sub-lists per kind of flagged tile (one per lane of the fallback kernel's scan)
constexpr unsigned FB_PAD = 32;    // ints between two sub-list counters: one 128-byte line each
constexpr unsigned FB_KINDS = 4;   // 0 literal chain, 1 constant, 2 rank 1, 3 one small singular value (B kept in fb_b)

// The iteration every tile kernel spends its time in (B = X V by one-sided Jacobi) is a generated,
// hand-scheduled gfx950 stream with pinned registers (tools/gen_jacobi_asm.py); -DWM_NO_ASM_JACOBI
// builds the C++ form of wm_tile_math.h instead (A/B, and what the CPU harness tests).
#if !defined(WM_NO_ASM_JACOBI)
#include "wm_jacobi_gfx950.inc"
#endif

// singular values of a raw tile (sigma-only kernels: K2, extract, detect); < 0: sweep bound hit
__device__ __forceinline__ int sigma_tile_dev(const wm::RawTile& raw, float (&s)[8]) {
#if !defined(WM_NO_ASM_JACOBI)
  wm::v2f a[4][8];
  float n2[8];
  // sigma only: test from the 3rd sweep on at cos^2 <= 1e-3 (the values are an order ahead of the vectors); from the
  // same sweep on a pair that is below cos^2 = 1e-8 in every tile of the wave is left alone (a third of the 4th
  // sweep's pair visits, tools/skip_study.cpp).  The threshold is set by the near-degenerate pairs: a residual
  // cosine c between two equal singular values moves them by c s_i / 2, so 1e-8 bounds the error at 5e-5 s_i;
  // 1e-6 measured 4.7e-4 s_1 on one tile of a 4K noise frame (profiles/r02m_sigma_skip.log), 1e-7 and 1e-8 leave
  // every value of four 4K frames where the full sweeps leave it (5e-6 s_1).
#ifndef WM_SIGMA_SKIP2
#define WM_SIGMA_SKIP2 1e-8f
#endif
#ifndef WM_SIGMA_SKIP_FROM
#define WM_SIGMA_SKIP_FROM 2
#endif
  const unsigned long long more = jacobi_cols_gfx950(raw.lo, raw.hi, a, n2, wm::JAC_CONV2_SIGMA, WM_SIGMA_SKIP2, 3, WM_SIGMA_SKIP_FROM);
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = wm::fsqrt(n2[i]);
  return more ? -1 : 1;
#else
  return wm::sigma_tile_pk(raw, s);
#endif
}

// ---------------------------------------------------------------------------
// tile I/O (per lane)
// ---------------------------------------------------------------------------
template <bool ALIGNED>
__device__ __forceinline__ void load_tile_u8(const uint8_t* __restrict__ p, const size_t stride,
                                             float (&a)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t lo, hi;
    if (ALIGNED) {
      const uint2 w = *reinterpret_cast<const uint2*>(p + r * stride);
      lo = w.x; hi = w.y;
    } else {
      const uint8_t* q = p + r * stride;
      lo = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
      hi = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
    }
    a[r][0] = (float)(lo & 0xffu); a[r][1] = (float)((lo >> 8) & 0xffu);
    a[r][2] = (float)((lo >> 16) & 0xffu); a[r][3] = (float)(lo >> 24);
    a[r][4] = (float)(hi & 0xffu); a[r][5] = (float)((hi >> 8) & 0xffu);
    a[r][6] = (float)((hi >> 16) & 0xffu); a[r][7] = (float)(hi >> 24);
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void store_tile_u8(uint8_t* __restrict__ p, const size_t stride,
                                              const float (&a)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t lo = wm::quant_u8(a[r][0]) | (wm::quant_u8(a[r][1]) << 8) |
                        (wm::quant_u8(a[r][2]) << 16) | (wm::quant_u8(a[r][3]) << 24);
    const uint32_t hi = wm::quant_u8(a[r][4]) | (wm::quant_u8(a[r][5]) << 8) |
                        (wm::quant_u8(a[r][6]) << 16) | (wm::quant_u8(a[r][7]) << 24);
    if (ALIGNED) {
      *reinterpret_cast<uint2*>(p + r * stride) = make_uint2(lo, hi);
    } else {
      uint8_t* q = p + r * stride;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q[i] = (uint8_t)(lo >> (8 * i));
        q[4 + i] = (uint8_t)(hi >> (8 * i));
      }
    }
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void load_raw(const uint8_t* __restrict__ p, const size_t stride,
                                         wm::RawTile& t) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (ALIGNED) {
      const uint2 w = *reinterpret_cast<const uint2*>(p + r * stride);
      t.lo[r] = w.x; t.hi[r] = w.y;
    } else {
      const uint8_t* q = p + r * stride;
      t.lo[r] = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
      t.hi[r] = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
    }
  }
}

// one 4-pixel word per row (half a tile)
template <bool ALIGNED>
__device__ __forceinline__ void load_words(const uint8_t* __restrict__ p, const size_t stride, uint32_t (&w)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint8_t* q = p + r * stride;
    if (ALIGNED) w[r] = *reinterpret_cast<const uint32_t*>(q);
    else w[r] = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
  }
}
template <bool ALIGNED>
__device__ __forceinline__ void store_words(uint8_t* __restrict__ p, const size_t stride, const uint32_t (&w)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint8_t* q = p + r * stride;
    if (ALIGNED) *reinterpret_cast<uint32_t*>(q) = w[r];
    else {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = (uint8_t)(w[r] >> (8 * i));
    }
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void store_raw(uint8_t* __restrict__ p, const size_t stride,
                                          const wm::RawTile& t) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (ALIGNED) {
      *reinterpret_cast<uint2*>(p + r * stride) = make_uint2(t.lo[r], t.hi[r]);
    } else {
      uint8_t* q = p + r * stride;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q[i] = (uint8_t)(t.lo[r] >> (8 * i));
        q[4 + i] = (uint8_t)(t.hi[r] >> (8 * i));
      }
    }
  }
}

template <bool VEC>
__device__ __forceinline__ void load_row8_f32(const float* __restrict__ p, float (&row)[8]) {
  if (VEC) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    const float4 y = *reinterpret_cast<const float4*>(p + 4);
    row[0] = x.x; row[1] = x.y; row[2] = x.z; row[3] = x.w;
    row[4] = y.x; row[5] = y.y; row[6] = y.z; row[7] = y.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) row[i] = p[i];
  }
}

template <bool VEC>
__device__ __forceinline__ void store_row8_f32(float* __restrict__ p, const float (&row)[8]) {
  if (VEC) {
    *reinterpret_cast<float4*>(p) = make_float4(row[0], row[1], row[2], row[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(row[4], row[5], row[6], row[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = row[i];
  }
}

// 8x8 float matrix stored contiguously (tile-major meta arrays: 256 B per tile)
__device__ __forceinline__ void load_mat_f32(const float* __restrict__ p, float (&m)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) load_row8_f32<true>(p + r * 8, m[r]);
}
__device__ __forceinline__ void store_mat_f32(float* __restrict__ p, const float (&m)[8][8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) store_row8_f32<true>(p + r * 8, m[r]);
}

struct Geom {
  int nbx;          // tiles per tile-row
  int n_tiles;      // tiles per plane
  int W;            // plane width (dense float outputs use W as row stride)
  size_t HW;        // H * W
  size_t row_stride;
  size_t plane_stride;
};

__device__ __forceinline__ bool tile_coords(const Geom& g, int& t, int& ty, int& tx) {
  t = blockIdx.x * WAVE + threadIdx.x;
  if (t >= g.n_tiles) return false;
  ty = t / g.nbx;
  tx = t - ty * g.nbx;
  return true;
}

// Plane-fastest, XCD-aware order for kernels whose planes SHARE per-tile inputs (the frames of a
// clip share Ux / Vxt: 512 B per tile).  Workgroups are dealt round-robin over the 8 XCDs, so
// b and b + 8 share an L2 (observed placement - it only affects speed, never results).  XCD x
// walks tile groups x, x + 8, ... and, for each group, all planes back to back: the group's
// shared inputs come from HBM once and are served to the other planes by that XCD's L2, instead
// of once per plane (grid (groups, planes): 66 MB of factors re-fetched 32 times at 32 x 4K).
constexpr unsigned N_XCD = 8;
__device__ __forceinline__ bool tile_coords_planefast(const Geom& g, const unsigned n_planes, int& t, int& ty,
                                                      int& tx, size_t& plane) {
#if defined(WM_EXP_EXTRACT_OLDMAP)   // A/B only: the round-1 order (all tile groups of plane 0, then plane 1, ...)
  const unsigned per_plane = gridDim.x / n_planes;
  const unsigned grp = blockIdx.x % per_plane;
  plane = blockIdx.x / per_plane;
#else
  const unsigned b = blockIdx.x, x = b % N_XCD, k = b / N_XCD;
  const unsigned grp = x + N_XCD * (k / n_planes);
  plane = k % n_planes;
#endif
  t = (int)(grp * WAVE + threadIdx.x);
  if (t >= g.n_tiles) return false;
  ty = t / g.nbx;
  tx = t - ty * g.nbx;
  return true;
}
inline dim3 tile_grid_planefast(const Geom& g, int n_planes) {
  const size_t groups = ((size_t)g.n_tiles + WAVE - 1) / WAVE;
  const size_t per_xcd = (groups + N_XCD - 1) / N_XCD;
  return dim3((unsigned)(per_xcd * N_XCD * (size_t)n_planes), 1, 1);
}

// ---------------------------------------------------------------------------
// K1  fused embed   (a1 a2 a3 a4 a5 a6 a7; sigma_c side output)
// ---------------------------------------------------------------------------
// Fast path: packed, V-free, pixel-domain (wm_tile_math.h identities (1),(2)).
// Flat / rank-deficient tiles append their id to one of the flagged-tile lists in `fb_list` (see append_kind)
// and are redone by k_embed_fallback.
#ifndef WM_EMBED_WAVES
#define WM_EMBED_WAVES 3
#endif
template <bool ALIGNED, bool YW>
__device__ __forceinline__ void embed_group(
    const uint8_t* host, const float* __restrict__ sigma_w,
    uint8_t* stego, float* __restrict__ sigma_c, float* __restrict__ yw,
    const Geom& g, const size_t sw_plane_stride, const float alpha, const int K,
    int* __restrict__ status, uint32_t* __restrict__ fb_list, int* __restrict__ fb_cnt, const uint32_t fb_cap,
    float* __restrict__ fb_b, const uint32_t fb_cap3,
    const int t, const int ty, const int tx, const size_t plane) {
  const size_t off = plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;

  // Flagged tiles are appended (wave-aggregated: one atomic per wave and kind) to one of three lists - 0: the literal
  // chain, 1: constant tiles, 2: rank-1 tiles - each split into FB_SUB sub-lists with their own counters, a wave using
  // sub-list (wave id % FB_SUB).  One counter per kind serialised the whole launch on content made of such tiles: 16 200
  // same-address atomics at ~12 ns each were 189 of the fast kernel's 189 us on an all-flat 8 x 4K batch (round 3,
  // gpurun_out/r03o).  Sub-list s of kind k: fb_list[(k * FB_SUB + s) * fb_cap ...], counter fb_cnt[(k * FB_SUB + s) * FB_PAD].
  const unsigned fb_sub = blockIdx.x % FB_SUB;
  auto append_kind = [&](const bool flag, const int kind) {
    const unsigned long long dmask = __builtin_amdgcn_ballot_w64(flag);
    if (dmask == 0ull) return;
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned leader = (unsigned)__builtin_ctzll(dmask);
    const unsigned l = (unsigned)kind * FB_SUB + fb_sub;
    int base = 0;
    if (lane == leader) base = atomicAdd(fb_cnt + (size_t)l * FB_PAD, (int)__builtin_popcountll(dmask));
    base = __builtin_amdgcn_readlane(base, leader);
    if (flag) {
      const unsigned rank = (unsigned)__builtin_popcountll(dmask & ((1ull << lane) - 1ull));
      fb_list[(size_t)l * fb_cap + base + rank] = (uint32_t)(plane * g.n_tiles + t);
    }
  };
  wm::v2f a[4][8];
  float n2[8];
  int sweeps;
  bool cst;
  int r1;
  {
    wm::RawTile raw;
    load_raw<ALIGNED>(host + off, g.row_stride, raw);
    // A wave of constant tiles only (letterbox bars, flat backgrounds) has nothing to iterate on: all of them are
    // rank-deficient and go to the constant list, which k_embed_fallback finishes in closed form
    // (wm::embed_tile_constant).  A constant tile in a mixed wave rides the iteration along and goes to the same list.
    // (the 16-word comparison only if some tile of the wave passes a two-word test: textured content pays 3 instructions)
    cst = false;
    if (__builtin_amdgcn_ballot_w64(raw.lo[0] == raw.hi[0] && raw.lo[0] == raw.hi[7]) != 0ull) cst = wm::raw_is_constant(raw);
    // rank-1 tiles that are not constant (edges of flat rectangles, rules and their crossings): closed form as well
    // (wm::embed_tile_rank1, third list).  The exact test (64 integer products) only runs if some tile of the wave passes
    // three 2x2 minors: textured waves pay ~15 instructions.
    r1 = 0;
    if (__builtin_amdgcn_ballot_w64(!cst && wm::raw_rank1_pretest(raw)) != 0ull && !cst && wm::raw_is_rank1(raw)) r1 = 1;
    if (__builtin_amdgcn_ballot_w64(!cst && r1 == 0) == 0ull) {      // nothing in this wave needs the iteration
      append_kind(cst, 1);
      append_kind(r1 != 0, 2);
      return;
    }
#if !defined(WM_NO_ASM_JACOBI)
    // sweeps 1-3 run untested, the 4th is the first that can be the last, pairs are skipped from the 5th on
    sweeps = jacobi_cols_gfx950(raw.lo, raw.hi, a, n2, wm::JAC_CONV2, wm::JAC_SKIP2, 4, 4) ? -1 : 1;
#else
    sweeps = wm::embed_jacobi_pk(raw, a, n2);
#endif
  }
  // Only B (64 VGPRs) crosses the sweep loop; everything else is (re)loaded when it is used
  // and stored as soon as it is final, so that the kernel fits 128 VGPRs (4 waves per SIMD):
  // watermark sigma -> coefficients -> Sc out; then the two 4-column halves of the tile one
  // after the other, each re-reading its row words (L2) and writing its stego words.
  asm volatile("" ::: "memory");
  float e[8];
  bool deficient;
  {
    float sw[8], sc[8], alpha_k[8];
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
#pragma unroll
    for (int i = 0; i < 8; ++i) alpha_k[i] = (i < K) ? alpha : 0.0f;
    wm::embed_coeffs_pk(n2, sw, alpha_k, e, sc, deficient);
    deficient = deficient || cst || r1 != 0;   // (constant and rank-1 tiles always are: sigma_8 = 0)
    // flagged tiles are left untouched: stego may alias host (in-place embedding) and the
    // fallback kernel must still read the original pixels; it also writes their Sc
    if (!deficient) store_row8_f32<true>(sigma_c + (plane * g.n_tiles + t) * 8, sc);
  }
  // Every other flagged tile - one singular value out of reach (noise / camera content) or rank 2 .. 6 (structured content) -
  // is kind 3: k_embed_one_small completes it from THIS B (wm::embed_tile_one_small / embed_tile_from_b), no Jacobi with
  // V.  B goes to fb_b (256 bytes per tile; kind 3's sub-lists hold fb_cap3 entries each, what does not fit takes the
  // literal chain).
  bool one_small = deficient && !cst && r1 == 0;
#if defined(WM_EXP_NO_ONE_SMALL)    // A/B only: everything flagged that is not constant / rank 1 takes the literal chain (round 2 behaviour)
  one_small = false;
#endif
  {
    const unsigned long long dmask = __builtin_amdgcn_ballot_w64(one_small);
    if (dmask != 0ull) {
      const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const unsigned leader = (unsigned)__builtin_ctzll(dmask);
      const unsigned l = 3u * FB_SUB + fb_sub;
      int base = 0;
      if (lane == leader) base = atomicAdd(fb_cnt + (size_t)l * FB_PAD, (int)__builtin_popcountll(dmask));
      base = __builtin_amdgcn_readlane(base, leader);
      if (one_small) {
        const unsigned slot = (unsigned)base + (unsigned)__builtin_popcountll(dmask & ((1ull << lane) - 1ull));
        if (slot < fb_cap3) {
          const size_t pos = (size_t)fb_sub * fb_cap3 + slot;
          fb_list[(size_t)3 * FB_SUB * fb_cap + pos] = (uint32_t)(plane * g.n_tiles + t);
          wm::v2f* dstb = reinterpret_cast<wm::v2f*>(fb_b + pos * 64);
#pragma unroll
          for (int rp = 0; rp < 4; ++rp)
#pragma unroll
            for (int c = 0; c < 8; ++c) dstb[rp * 8 + c] = a[rp][c];
        } else {
          one_small = false;                        // sub-list full: the literal chain
        }
      }
    }
  }
  append_kind(deficient && !cst && r1 == 0 && !one_small, 0);
  append_kind(cst, 1);
  append_kind(r1 != 0, 2);
  if (sweeps < 0) atomicOr(status, 1);
  float* ywp = YW ? yw + plane * g.HW + (size_t)ty * 8 * g.W + (size_t)tx * 8 : nullptr;
#pragma nounroll
  for (int half = 0; half < 2; ++half) {      // a real loop: one half's registers at a time
    uint32_t w[8], ow[8];
    load_words<ALIGNED>(host + off + 4 * half, g.row_stride, w);
    wm::embed_half_pk<YW>(w, a, e, ow, YW ? ywp + 4 * half : nullptr, (size_t)g.W);
    if (!deficient) store_words<ALIGNED>(stego + off + 4 * half, g.row_stride, ow);
    asm volatile("" ::: "memory");
  }
}

// One wave per workgroup and one workgroup per (tile group, plane): 64 800 workgroups per
// 32 x 4K launch.  A persistent grid (CUs x resident waves, grid-stride over the work) was
// measured 4 % SLOWER in the same process (profiles/r02_embed_variants.md): the dispatcher's
// dynamic placement balances waves that need 4 against waves that need 5 sweeps, a fixed
// stride does not.
template <bool ALIGNED, bool YW>
__global__ __launch_bounds__(WAVE, WM_EMBED_WAVES) void k_embed_tiles(
    const uint8_t* host, const float* __restrict__ sigma_w,
    uint8_t* stego, float* __restrict__ sigma_c, float* __restrict__ yw,
    const Geom g, const unsigned n_groups, const size_t sw_plane_stride,
    const float alpha, const int K, int* __restrict__ status, uint32_t* __restrict__ fb_list, int* __restrict__ fb_cnt,
    const uint32_t fb_cap, float* __restrict__ fb_b, const uint32_t fb_cap3) {
  const unsigned w = blockIdx.x;
  const unsigned plane = w / n_groups, grp = w - plane * n_groups;
  const int t = (int)(grp * WAVE + threadIdx.x);
  if (t >= g.n_tiles) return;
  const int ty = t / g.nbx, tx = t - ty * g.nbx;
  embed_group<ALIGNED, YW>(host, sigma_w, stego, sigma_c, yw, g, sw_plane_stride, alpha, K, status, fb_list, fb_cnt, fb_cap, fb_b, fb_cap3,
                           t, ty, tx, (size_t)plane);
}

// The tiles listed by the fast kernel, one per lane, a fixed grid striding each list (the counts are only known on
// the device): first the literal chain with orthonormal completion (wm::embed_tile_completed) for the front list,
// then the closed form of the constant tiles (wm::embed_tile_constant) for the back list.
#ifndef WM_FALLBACK_WAVES
#define WM_FALLBACK_WAVES 2
#endif
template <bool ALIGNED, bool YW>
__global__ __launch_bounds__(WAVE, WM_FALLBACK_WAVES) void k_embed_fallback(
    const uint8_t* host, const float* __restrict__ sigma_w,
    uint8_t* stego, float* __restrict__ sigma_c, float* __restrict__ yw,
    const Geom g, const size_t sw_plane_stride, const float alpha, const int K,
    int* __restrict__ status, const uint32_t* __restrict__ fb_list, const int* __restrict__ fb_cnt, const uint32_t fb_cap,
    const float* __restrict__ fb_b, const uint32_t fb_cap3) {
  float alpha_k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) alpha_k[i] = (i < K) ? alpha : 0.0f;
  auto finish = [&](const size_t plane, const int t, const size_t off, const float (&sc)[8], float (&a)[8][8]) {
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    store_row8_f32<true>(sigma_c + (plane * g.n_tiles + t) * 8, sc);
    store_tile_u8<ALIGNED>(stego + off, g.row_stride, a);
    if (YW) {
      float* o = yw + plane * g.HW + (size_t)ty * 8 * g.W + (size_t)tx * 8;
#pragma unroll
      for (int r = 0; r < 8; ++r) store_row8_f32<false>(o + (size_t)r * g.W, a[r]);
    }
  };
  // item `it` of kind k: the sub-lists' counts are scanned in the wave (lane l holds sub-list l), the owning sub-list is
  // found by a 6-step search over the lanes' prefix sums
  static_assert(FB_SUB == WAVE, "one sub-list per lane");
  int pre[FB_KINDS], total[FB_KINDS];
#pragma unroll
  for (int k = 0; k < (int)FB_KINDS; ++k) {
    int c = fb_cnt[(size_t)(k * FB_SUB + threadIdx.x) * FB_PAD];
    if (k == 3) c = min(c, (int)fb_cap3);           // what did not fit went to kind 0
    int incl = c;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int v = __shfl_up(incl, o, WAVE); if ((int)threadIdx.x >= o) incl += v; }
    pre[k] = incl - c;
    total[k] = __shfl(incl, WAVE - 1, WAVE);
  }
  auto locate = [&](const int k, const int it, int& sub, int& j) {      // `it` < total[k]
    int s_ = 0;
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) { const int p_ = __shfl(pre[k], s_ + o, WAVE); if (p_ <= it) s_ += o; }
    sub = s_; j = it - __shfl(pre[k], s_, WAVE);
  };
  auto item = [&](const int k, const int it) -> uint32_t {
    int sub, j;
    locate(k, it, sub, j);
    return fb_list[(size_t)(k * FB_SUB + sub) * fb_cap + j];
  };
  const int count = total[0], n_const = total[1], n_rank1 = total[2];
  for (int it0 = blockIdx.x * WAVE; it0 < count; it0 += gridDim.x * WAVE) {      // wave-uniform bounds: the shuffles need every lane
    const int it = it0 + threadIdx.x;
    const uint32_t id = item(0, min(it, count - 1));
    if (it >= count) continue;
    const size_t plane = id / (uint32_t)g.n_tiles;
    const int t = (int)(id % (uint32_t)g.n_tiles);
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    const size_t off = plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;
    float a[8][8], sw[8], sc[8];
    load_tile_u8<ALIGNED>(host + off, g.row_stride, a);
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
    if (wm::embed_tile_completed(a, sw, alpha_k, sc) < 0) atomicOr(status, 1);
    finish(plane, t, off, sc, a);
  }
  for (int it0 = blockIdx.x * WAVE; it0 < n_const; it0 += gridDim.x * WAVE) {
    const int it = it0 + threadIdx.x;
    const uint32_t id = item(1, min(it, n_const - 1));
    if (it >= n_const) continue;
    const size_t plane = id / (uint32_t)g.n_tiles;
    const int t = (int)(id % (uint32_t)g.n_tiles);
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    const size_t off = plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;
    float a[8][8], sw[8], sc[8];
    const float v0 = (float)host[off];                    // every pixel of the tile has this value
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
    // the table is a compile-time constant when the wave's tiles are all black or none of them is (bit-identical
    // to the general form, which adds 0 * the other table)
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(v0 == 0.0f);
    if (bm == 0ull) wm::embed_tile_constant_t<1>(v0, sw, alpha_k, sc, a);
    else if (bm == __builtin_amdgcn_ballot_w64(true)) wm::embed_tile_constant_t<0>(v0, sw, alpha_k, sc, a);
    else wm::embed_tile_constant(v0, sw, alpha_k, sc, a);
    finish(plane, t, off, sc, a);
  }
  for (int it0 = blockIdx.x * WAVE; it0 < n_rank1; it0 += gridDim.x * WAVE) {
    const int it = it0 + threadIdx.x;
    const uint32_t id = item(2, min(it, n_rank1 - 1));
    if (it >= n_rank1) continue;
    const size_t plane = id / (uint32_t)g.n_tiles;
    const int t = (int)(id % (uint32_t)g.n_tiles);
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    const size_t off = plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;
    float a[8][8], sw[8], sc[8];
    load_tile_u8<ALIGNED>(host + off, g.row_stride, a);
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
    wm::embed_tile_rank1(a, sw, alpha_k, sc, a);
    finish(plane, t, off, sc, a);
  }
}

// Kind 3 of the flagged-tile lists (completed from the fast kernel's B: wm::embed_tile_one_small / embed_tile_from_b) in a
// kernel of its own, one wave per SIMD: three 8 x 8 arrays and a float64 bilinear form need 314-362 registers; inside
// k_embed_fallback they pushed the literal chain's 202-228 VGPRs into scratch.  Same list walk as k_embed_fallback (one
// tile per lane, a fixed grid striding the list - the count is only known on the device).
template <bool ALIGNED, bool YW>
__global__ __launch_bounds__(WAVE, 1) void k_embed_one_small(
    const uint8_t* host, const float* __restrict__ sigma_w,
    uint8_t* stego, float* __restrict__ sigma_c, float* __restrict__ yw,
    const Geom g, const size_t sw_plane_stride, const float alpha, const int K,
    int* __restrict__ status, const uint32_t* __restrict__ fb_list, const int* __restrict__ fb_cnt, const uint32_t fb_cap,
    const float* __restrict__ fb_b, const uint32_t fb_cap3) {
  float alpha_k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) alpha_k[i] = (i < K) ? alpha : 0.0f;
  auto finish = [&](const size_t plane, const int t, const size_t off, const float (&sc)[8], float (&a)[8][8]) {
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    store_row8_f32<true>(sigma_c + (plane * g.n_tiles + t) * 8, sc);
    store_tile_u8<ALIGNED>(stego + off, g.row_stride, a);
    if (YW) {
      float* o = yw + plane * g.HW + (size_t)ty * 8 * g.W + (size_t)tx * 8;
#pragma unroll
      for (int r = 0; r < 8; ++r) store_row8_f32<false>(o + (size_t)r * g.W, a[r]);
    }
  };
  // item `it` of kind k: the sub-lists' counts are scanned in the wave (lane l holds sub-list l), the owning sub-list is
  // found by a 6-step search over the lanes' prefix sums
  static_assert(FB_SUB == WAVE, "one sub-list per lane");
  int pre[FB_KINDS], total[FB_KINDS];
#pragma unroll
  for (int k = 0; k < (int)FB_KINDS; ++k) {
    int c = fb_cnt[(size_t)(k * FB_SUB + threadIdx.x) * FB_PAD];
    if (k == 3) c = min(c, (int)fb_cap3);           // what did not fit went to kind 0
    int incl = c;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int v = __shfl_up(incl, o, WAVE); if ((int)threadIdx.x >= o) incl += v; }
    pre[k] = incl - c;
    total[k] = __shfl(incl, WAVE - 1, WAVE);
  }
  auto locate = [&](const int k, const int it, int& sub, int& j) {      // `it` < total[k]
    int s_ = 0;
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) { const int p_ = __shfl(pre[k], s_ + o, WAVE); if (p_ <= it) s_ += o; }
    sub = s_; j = it - __shfl(pre[k], s_, WAVE);
  };
  auto item = [&](const int k, const int it) -> uint32_t {
    int sub, j;
    locate(k, it, sub, j);
    return fb_list[(size_t)(k * FB_SUB + sub) * fb_cap + j];
  };
  const int n_small = total[3];
  for (int it0 = blockIdx.x * WAVE; it0 < n_small; it0 += gridDim.x * WAVE) {
    const int it = it0 + threadIdx.x;
    int sub, j;
    locate(3, min(it, n_small - 1), sub, j);
    if (it >= n_small) continue;
    const size_t pos = (size_t)sub * fb_cap3 + j;
    const uint32_t id = fb_list[(size_t)3 * FB_SUB * fb_cap + pos];
    const size_t plane = id / (uint32_t)g.n_tiles;
    const int t = (int)(id % (uint32_t)g.n_tiles);
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    const size_t off = plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;
    float a[8][8], bb[8][8], sw[8], sc[8];
    load_tile_u8<ALIGNED>(host + off, g.row_stride, a);
    const wm::v2f* srcb = reinterpret_cast<const wm::v2f*>(fb_b + pos * 64);
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
#pragma unroll
      for (int c = 0; c < 8; ++c) { const wm::v2f v = srcb[rp * 8 + c]; bb[2 * rp][c] = v[0]; bb[2 * rp + 1][c] = v[1]; }
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
    float n6 = 0.0f, n0 = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { n6 = __builtin_fmaf(bb[r][6], bb[r][6], n6); n0 = __builtin_fmaf(bb[r][0], bb[r][0], n0); }
    // one missing pair: its joint sign is defined (float64 inside); more than one: any orthonormal completion
    const bool one = n6 > wm::SIGMA_RATIO_MIN2 * n0;
    if (wm::wave_any(one)) { if (one) wm::embed_tile_one_small(a, bb, sw, alpha_k, sc, a); }
    if (wm::wave_any(!one)) { if (!one) wm::embed_tile_from_b(a, bb, sw, alpha_k, sc, a); }
    finish(plane, t, off, sc, a);
  }
}

// copy the rows/columns no tile covers (H % 8, W % 8) from host to stego,
// and into yw as float
__global__ void k_copy_border(const uint8_t* host, uint8_t* stego,
                              float* __restrict__ yw, const int H, const int W, const int Hb,
                              const int Wb, const size_t row_stride, const size_t plane_stride) {
  const size_t plane = blockIdx.y;
  const int n_right = (W - Wb) * Hb;            // right strip: rows [0,Hb) cols [Wb,W)
  const int n_bottom = (H - Hb) * W;            // bottom strip: rows [Hb,H) all cols
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_right + n_bottom;
       i += gridDim.x * blockDim.x) {
    int r, c;
    if (i < n_right) { r = i / (W - Wb); c = Wb + i % (W - Wb); }
    else { const int j = i - n_right; r = Hb + j / W; c = j % W; }
    const size_t o = plane * plane_stride + (size_t)r * row_stride + c;
    const uint8_t px = host[o];
    if (stego != host) stego[o] = px;
    if (yw) yw[plane * (size_t)H * W + (size_t)r * W + c] = (float)px;
  }
}

// ---------------------------------------------------------------------------
// K2  sigma only
// ---------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(WAVE, 3) void k_sigma_tiles(const uint8_t* __restrict__ planes,
                                                     float* __restrict__ sigma, const Geom g,
                                                     int* __restrict__ status) {
  int t, ty, tx;
  if (!tile_coords(g, t, ty, tx)) return;
  const size_t plane = blockIdx.y;
  wm::RawTile raw;
  float s[8];
  load_raw<ALIGNED>(planes + plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8,
                    g.row_stride, raw);
  if (sigma_tile_dev(raw, s) < 0) atomicOr(status, 1);
  store_row8_f32<true>(sigma + (plane * g.n_tiles + t) * 8, s);
}

// ---------------------------------------------------------------------------
// K3  full SVD of float tiles (watermark side)
// ---------------------------------------------------------------------------
template <bool VECF>
__global__ __launch_bounds__(WAVE, 2) void k_svd_tiles(const float* __restrict__ planes,
                                                   float* __restrict__ U, float* __restrict__ S,
                                                   float* __restrict__ Vt, const Geom g,
                                                   int* __restrict__ status) {
  int t, ty, tx;
  if (!tile_coords(g, t, ty, tx)) return;
  const size_t plane = blockIdx.y;
  const float* p = planes + plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8;
  float a[8][8], s[8], vt[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) load_row8_f32<VECF>(p + (size_t)r * g.row_stride, a[r]);
  if (wm::svd_tile(a, s, vt) < 0) atomicOr(status, 1);
  const size_t ti = plane * g.n_tiles + t;
  const bool deficient = !(s[7] > 1e-5f * s[0]);
  if (!deficient) {
    store_row8_f32<true>(S + ti * 8, s);
    store_mat_f32(U + ti * 64, a);
    store_mat_f32(Vt + ti * 64, vt);
  }
  // rank-deficient watermark tiles: redo with the completion pattern so that Uw
  // stays a full orthonormal basis (rare: wave-uniform branch, nothing kept live across it)
  if (wm::wave_any(deficient)) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) load_row8_f32<VECF>(p + (size_t)r * g.row_stride, a[r]);
    if (wm::svd_tile(a, s, vt, true) < 0) atomicOr(status, 1);
    if (deficient) {
      store_row8_f32<true>(S + ti * 8, s);
      store_mat_f32(U + ti * 64, a);
      store_mat_f32(Vt + ti * 64, vt);
    }
  }
}

// ---------------------------------------------------------------------------
// K2+K4  fused extract
// ---------------------------------------------------------------------------
// MM: the wave also leaves the {min, max} of its 4 096 outputs (order-preserving uint form) in mm[plane][tile group]: the
// min-max pass of the normalise that follows the unscramble (single:221) costs nothing extra then.  Lanes past the last tile
// of a partial wave recompute the last tile (every lane takes part in the reduction) and store nothing.
template <bool ALIGNED, bool VECF, bool PX, bool MM>
__global__ __launch_bounds__(WAVE, 3) void k_extract_tiles(
    const uint8_t* __restrict__ stego, const float* __restrict__ sigma_c,
    const float* __restrict__ Uw, const float* __restrict__ Vwt, float* __restrict__ out,
    const Geom g, const unsigned n_planes, const size_t uv_plane_stride, const float inv_alpha, const int K,
    int* __restrict__ status, unsigned* __restrict__ mm) {
  int t, ty, tx;
  size_t plane;
  bool valid = true;
  if (MM) {
    const unsigned b = blockIdx.x, x = b % N_XCD, k = b / N_XCD;
    const unsigned grp = x + N_XCD * (k / n_planes);
    plane = k % n_planes;
    if ((size_t)grp * WAVE >= (size_t)g.n_tiles) return;               // wave-uniform: a padding group
    t = (int)(grp * WAVE + threadIdx.x);
    valid = t < g.n_tiles;
    t = valid ? t : g.n_tiles - 1;
    ty = t / g.nbx; tx = t - ty * g.nbx;
  } else if (!tile_coords_planefast(g, n_planes, t, ty, tx, plane)) return;
  wm::RawTile raw;
  float a[8][8], s[8], sc[8], keep[8];
  load_raw<ALIGNED>(stego + plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8,
                    g.row_stride, raw);
  if (sigma_tile_dev(raw, s) < 0) atomicOr(status, 1);
  load_row8_f32<true>(sigma_c + (plane * g.n_tiles + t) * 8, sc);
#pragma unroll
  for (int i = 0; i < 8; ++i) keep[i] = (i < K) ? 1.0f : 0.0f;
  float uw[8][8], vwt[8][8];
  const size_t mi = (plane * uv_plane_stride + (size_t)t) * 64;
  load_mat_f32(Uw + mi, uw);
  load_mat_f32(Vwt + mi, vwt);
  if (PX) wm::extract_tile_px(s, sc, inv_alpha, keep, uw, vwt, a);
  else wm::extract_tile(s, sc, inv_alpha, keep, uw, vwt, a);
  float* o = out + plane * g.HW + (size_t)ty * 8 * g.W + (size_t)tx * 8;
  if (valid) {
#pragma unroll
    for (int r = 0; r < 8; ++r) store_row8_f32<VECF>(o + (size_t)r * g.W, a[r]);
  }
  if (MM) {
    float lo = a[0][0], hi = a[0][0];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) { lo = fminf(lo, a[r][c]); hi = fmaxf(hi, a[r][c]); }
    unsigned ulo = f2ord(lo), uhi = f2ord(hi);
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) { ulo = min(ulo, (unsigned)__shfl_down(ulo, o_, WAVE)); uhi = max(uhi, (unsigned)__shfl_down(uhi, o_, WAVE)); }
    if (threadIdx.x == 0) {
      const unsigned b = blockIdx.x, grp = b % N_XCD + N_XCD * ((b / N_XCD) / n_planes);
      const size_t groups = ((size_t)g.n_tiles + WAVE - 1) / WAVE;
      mm[2 * (plane * groups + grp)] = ulo; mm[2 * (plane * groups + grp) + 1] = uhi;
    }
  }
}

// out[i] = sum over planes z (ascending, deterministic) of in[z][i]: the frames of a clip carry the
// same watermark, their estimates are averaged (video extract) - on the device, so that one plane
// instead of n crosses PCIe
__global__ __launch_bounds__(256) void k_sum_planes(const float* __restrict__ in, const size_t plane_elems, const int n,
                                                   float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane_elems; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.0f;
    for (int z = 0; z < n; ++z) acc += in[(size_t)z * plane_elems + i];
    out[i] = acc;
  }
}

// per-watermark preparation for the PX extract: Ux = D^T Uw, Vxt = Vwt D (in place allowed)
__global__ __launch_bounds__(WAVE) void k_factors_to_pixel(const float* Uw, const float* Vwt, float* Ux, float* Vxt,
                                                          const size_t n_tiles) {
  const size_t t = (size_t)blockIdx.x * WAVE + threadIdx.x;
  if (t >= n_tiles) return;
  float u[8][8], vt[8][8];
  load_mat_f32(Uw + t * 64, u);
  load_mat_f32(Vwt + t * 64, vt);
  wm::factors_to_pixel(u, vt);
  store_mat_f32(Ux + t * 64, u);
  store_mat_f32(Vxt + t * 64, vt);
}

// ---------------------------------------------------------------------------
// K4  reconstruct only:  Uw diag(sw_hat) Vwt -> idct
// ---------------------------------------------------------------------------
template <bool VECF>
__global__ __launch_bounds__(WAVE, 3) void k_reconstruct_tiles(
    const float* __restrict__ Uw, const float* __restrict__ sw_hat,
    const float* __restrict__ Vwt, float* __restrict__ out, const Geom g) {
  int t, ty, tx;
  if (!tile_coords(g, t, ty, tx)) return;
  const size_t plane = blockIdx.y;
  const size_t ti = plane * g.n_tiles + t;
  float sh[8], zero[8], one[8], uw[8][8], vwt[8][8], a[8][8];
  load_row8_f32<true>(sw_hat + ti * 8, sh);
#pragma unroll
  for (int i = 0; i < 8; ++i) { zero[i] = 0.0f; one[i] = 1.0f; }
  load_mat_f32(Uw + ti * 64, uw);
  load_mat_f32(Vwt + ti * 64, vwt);
  wm::extract_tile(sh, zero, 1.0f, one, uw, vwt, a);   // (sh - 0) * 1 * 1
  float* o = out + plane * g.HW + (size_t)ty * 8 * g.W + (size_t)tx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) store_row8_f32<VECF>(o + (size_t)r * g.W, a[r]);
}

// ---------------------------------------------------------------------------
// K2+K5  fused detect: per-wave partial sums, then one block per plane
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, WAVE);
  return x;
}

template <bool ALIGNED>
__global__ __launch_bounds__(WAVE, 3) void k_detect_tiles(
    const uint8_t* __restrict__ stego, const float* __restrict__ sigma_c,
    const float* __restrict__ sigma_w, double* __restrict__ partials, const Geom g,
    const size_t sw_plane_stride, const float inv_alpha, int* __restrict__ status) {
  const int t = blockIdx.x * WAVE + threadIdx.x;
  const size_t plane = blockIdx.y;
  double acc[N_SUMS] = {0, 0, 0, 0, 0};
  if (t < g.n_tiles) {
    const int ty = t / g.nbx, tx = t - ty * g.nbx;
    wm::RawTile raw;
    float s[8], sc[8], sw[8];
    load_raw<ALIGNED>(stego + plane * g.plane_stride + (size_t)ty * 8 * g.row_stride + (size_t)tx * 8,
                      g.row_stride, raw);
    if (sigma_tile_dev(raw, s) < 0) atomicOr(status, 1);
    load_row8_f32<true>(sigma_c + (plane * g.n_tiles + t) * 8, sc);
    load_row8_f32<true>(sigma_w + plane * sw_plane_stride + (size_t)t * 8, sw);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double x = (double)sw[i];
      const double y = (double)((s[i] - sc[i]) * inv_alpha);
      acc[0] += x; acc[1] += y; acc[2] += x * y; acc[3] += x * x; acc[4] += y * y;
    }
  }
#pragma unroll
  for (int k = 0; k < N_SUMS; ++k) acc[k] = wave_sum(acc[k]);
  if (threadIdx.x == 0) {
    double* o = partials + (plane * gridDim.x + blockIdx.x) * N_SUMS;
#pragma unroll
    for (int k = 0; k < N_SUMS; ++k) o[k] = acc[k];
  }
}

__global__ __launch_bounds__(256) void k_detect_finalize(const double* __restrict__ partials,
                                                        double* __restrict__ scores,
                                                        const int n_waves, const double n_vals) {
  __shared__ double sm[4][N_SUMS];
  const size_t plane = blockIdx.x;
  double acc[N_SUMS] = {0, 0, 0, 0, 0};
  for (int w = threadIdx.x; w < n_waves; w += blockDim.x) {
    const double* p = partials + (plane * n_waves + w) * N_SUMS;
#pragma unroll
    for (int k = 0; k < N_SUMS; ++k) acc[k] += p[k];
  }
#pragma unroll
  for (int k = 0; k < N_SUMS; ++k) acc[k] = wave_sum(acc[k]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N_SUMS; ++k) sm[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s[N_SUMS];
#pragma unroll
    for (int k = 0; k < N_SUMS; ++k) s[k] = sm[0][k] + sm[1][k] + sm[2][k] + sm[3][k];
    double score = 0.0;
    if (n_vals > 0) {
      // _nc (single:284-289): mean-removed correlation, +1e-8 in the denominator
      const double cov = s[2] - s[0] * s[1] / n_vals;
      double va = s[3] - s[0] * s[0] / n_vals;
      double vb = s[4] - s[1] * s[1] / n_vals;
      va = va > 0 ? va : 0;
      vb = vb > 0 ? vb : 0;
      score = cov / (sqrt(va) * sqrt(vb) + 1e-8);
    }
    scores[plane] = score;
  }
}

// ---------------------------------------------------------------------------
// host-side argument checking / launch helpers
// ---------------------------------------------------------------------------
int check_plane_args(const wm_ctx* ctx, const void* p, int n_planes, int H, int W, int row_stride,
                     size_t plane_stride) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || H < 0 || W < 0) return set_err(WM_ERR_BADARG, "negative size");
  if (!p && n_planes > 0 && H > 0 && W > 0) return set_err(WM_ERR_BADARG, "plane pointer is NULL");
  if (n_planes > 65535) return set_err(WM_ERR_BADARG, "n_planes > 65535");
  if (row_stride < W) return set_err(WM_ERR_BADARG, "row_stride < W");
  if (n_planes > 1 && plane_stride < (size_t)row_stride * (size_t)(H > 0 ? H - 1 : 0) + (size_t)W)
    return set_err(WM_ERR_BADARG, "plane_stride smaller than one plane");
  return WM_OK;
}

Geom make_geom(int H, int W, int row_stride, size_t plane_stride) {
  Geom g;
  g.nbx = W / 8;
  g.n_tiles = (H / 8) * (W / 8);
  g.W = W;
  g.HW = (size_t)H * (size_t)W;
  g.row_stride = (size_t)row_stride;
  g.plane_stride = plane_stride;
  return g;
}

inline bool u8_aligned(const void* a, const void* b, int row_stride, size_t plane_stride) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)row_stride | (uintptr_t)plane_stride) & 7u) == 0;
}
inline bool f32_vec_ok(const void* p, size_t row_stride_elems, size_t plane_stride_elems) {
  return (((uintptr_t)p) & 15u) == 0 && (row_stride_elems & 3u) == 0 && (plane_stride_elems & 3u) == 0;
}

inline dim3 tile_grid(const Geom& g, int n_planes) {
  return dim3((unsigned)((g.n_tiles + WAVE - 1) / WAVE), (unsigned)n_planes, 1);
}

}  // namespace

namespace {
struct Carve {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t plane_span(int n_planes, int H, int row_stride, size_t plane_stride, int W) {
  if (n_planes == 0 || H == 0) return 0;
  return (size_t)(n_planes - 1) * plane_stride + (size_t)(H - 1) * row_stride + (size_t)W;
}
}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int wm_abi_version(void) { return WM_ABI_VERSION; }
const char* wm_last_error(void) { return wmi::g_err; }

int wm_device_count(int* n_out) {
  if (!n_out) return set_err(WM_ERR_BADARG, "n_out is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  *n_out = n;
  return WM_OK;
}

int wm_create(int device, void* stream, wm_ctx** ctx_out) {
  if (!ctx_out) return set_err(WM_ERR_BADARG, "ctx_out is NULL");
  *ctx_out = nullptr;
  int n = 0;
  WM_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return set_err(WM_ERR_BADARG, "device index out of range");
  WM_HIP(hipSetDevice(device));
  wm_ctx* ctx = new (std::nothrow) wm_ctx();
  if (!ctx) return set_err(WM_ERR_NOMEM, "host allocation failed");
  ctx->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cu = cus;
    else (void)hipGetLastError();
  }
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return set_err(WM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    ctx->owns_stream = true;
  }
  hipError_t e = hipMalloc((void**)&ctx->d_status, 4 * sizeof(int));   // [0] sticky kernel status (the flagged-tile counters live behind fb_list)
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_status, 0, 4 * sizeof(int), ctx->stream);
  for (int i = 0; i < N_EVENTS && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
  if (e != hipSuccess) { wm_destroy(ctx); return set_err(WM_ERR_HIP, "context setup: %s", hipGetErrorString(e)); }
  *ctx_out = ctx;
  return WM_OK;
}

int wm_destroy(wm_ctx* ctx) {
  if (!ctx) return WM_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < N_EVENTS; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  for (int i = 0; i < wm_ctx::MAX_AUX; ++i) {
    if (ctx->aux_stream[i]) { (void)hipStreamSynchronize(ctx->aux_stream[i]); (void)hipStreamDestroy(ctx->aux_stream[i]); }
    if (ctx->ev_fork[i]) (void)hipEventDestroy(ctx->ev_fork[i]);
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->partials) (void)hipFree(ctx->partials);
  if (ctx->fb_list) (void)hipFree(ctx->fb_list);
  if (ctx->ref_ws) (void)hipFree(ctx->ref_ws);
  if (ctx->ref_ws2) (void)hipFree(ctx->ref_ws2);
  if (ctx->route_tmp) (void)hipFree(ctx->route_tmp);
  if (ctx->extract_f32) (void)hipFree(ctx->extract_f32);
  for (int i = 0; i < wm_ctx::MAX_PAIR_TABS; ++i) if (ctx->pair_tab[i]) (void)hipFree(ctx->pair_tab[i]);
  for (int i = 0; i < wm_ctx::MAX_PAIR_TABS; ++i) {
    if (ctx->hier_dev[i]) (void)hipFree(ctx->hier_dev[i]);
    if (ctx->hier_host[i]) wmi::hier_host_free(ctx->hier_host[i]);
  }
  if (ctx->hier_ws) (void)hipFree(ctx->hier_ws);
  for (int i = 0; i < 2; ++i) if (ctx->dct_mat[i]) (void)hipFree(ctx->dct_mat[i]);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return WM_OK;
}

int wm_sync(wm_ctx* ctx) {
  WM_TRY(wmi::use_ctx(ctx));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_check_status(wm_ctx* ctx) {
  WM_TRY(wmi::use_ctx(ctx));
  int st = 0;
  WM_HIP(hipMemcpyAsync(&st, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  if (st != 0) {
    WM_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
    return set_err(WM_ERR_NOCONV, "SVD did not converge");
  }
  return WM_OK;
}

int wm_malloc(wm_ctx* ctx, size_t bytes, void** dptr_out) {
  if (!ctx || !dptr_out) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_HIP(hipSetDevice(ctx->device));
  *dptr_out = nullptr;
  if (hipMalloc(dptr_out, bytes ? bytes : 1) != hipSuccess) {
    (void)hipGetLastError();
    return set_err(WM_ERR_NOMEM, "hipMalloc failed");
  }
  return WM_OK;
}

int wm_free(wm_ctx* ctx, void* dptr) {
  WM_TRY(wmi::use_ctx(ctx));
  if (dptr) { WM_HIP(hipStreamSynchronize(ctx->stream)); WM_HIP(hipFree(dptr)); }
  return WM_OK;
}

int wm_memcpy_h2d(wm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx || (bytes && (!dst_dev || !src_host))) return set_err(WM_ERR_BADARG, "NULL argument");
  if (bytes) WM_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return WM_OK;
}

int wm_memcpy_d2h(wm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx || (bytes && (!dst_host || !src_dev))) return set_err(WM_ERR_BADARG, "NULL argument");
  if (bytes) {
    WM_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WM_HIP(hipStreamSynchronize(ctx->stream));
  }
  return WM_OK;
}

int wm_memset(wm_ctx* ctx, void* dst_dev, int value, size_t bytes) {
  if (!ctx || (bytes && !dst_dev)) return set_err(WM_ERR_BADARG, "NULL argument");
  if (bytes) WM_HIP(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
  return WM_OK;
}

int wm_event_record(wm_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= N_EVENTS) return set_err(WM_ERR_BADARG, "bad event slot");
  WM_HIP(hipEventRecord(ctx->ev[slot], ctx->stream));
  return WM_OK;
}

int wm_event_elapsed_ms(wm_ctx* ctx, int slot_start, int slot_stop, float* ms_out) {
  if (!ctx || !ms_out || slot_start < 0 || slot_start >= N_EVENTS || slot_stop < 0 || slot_stop >= N_EVENTS)
    return set_err(WM_ERR_BADARG, "bad event slot");
  WM_HIP(hipEventSynchronize(ctx->ev[slot_stop]));
  WM_HIP(hipEventElapsedTime(ms_out, ctx->ev[slot_start], ctx->ev[slot_stop]));
  return WM_OK;
}

// ---- K1 --------------------------------------------------------------------
int wm_embed_tiles_u8_dev(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego,
                          float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                          size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K) {
  WM_TRY(check_plane_args(ctx, host, n_planes, H, W, row_stride, plane_stride));
  if (!stego) return set_err(WM_ERR_BADARG, "stego is NULL");
  if (K < 0 || K > 8) return set_err(WM_ERR_BADARG, "K must be in 0..8");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  if (g.n_tiles > 0) {
    if (!sigma_w || !sigma_c) return set_err(WM_ERR_BADARG, "sigma_w / sigma_c is NULL");
    if ((((uintptr_t)sigma_w | (uintptr_t)sigma_c) & 15u) || (sigma_w_plane_stride & 3u))
      return set_err(WM_ERR_BADARG, "sigma arrays must be 16-byte aligned");
    const bool al = u8_aligned(host, stego, row_stride, plane_stride);
    const dim3 block(WAVE);
    const unsigned n_groups = (unsigned)((g.n_tiles + WAVE - 1) / WAVE);
    const size_t n_work_sz = (size_t)n_groups * (size_t)n_planes;
    if (n_work_sz > 0x7fffffffull) return set_err(WM_ERR_BADARG, "more than 2^31 tile groups in one call");
    const dim3 grid((unsigned)n_work_sz);
    const size_t n_all = (size_t)g.n_tiles * (size_t)n_planes;
    if (n_all > 0x7fffffffull) return set_err(WM_ERR_BADARG, "more than 2^31 tiles in one call");   // ids are uint32, the device-side count an int
    const size_t n_waves = (n_all + WAVE - 1) / WAVE;
    // three kinds x FB_SUB sub-lists of fb_cap entries (a sub-list holds at most the tiles of the waves that hash to it),
    // then the padded counters
    const size_t cap = (n_work_sz + FB_SUB - 1) / FB_SUB * WAVE;
    if (cap > 0x7fffffffull) return set_err(WM_ERR_BADARG, "more than 2^31 tiles in one call");
    // kind 3 keeps its tiles' B (256 bytes each): its sub-lists hold cap / 16 entries (>= 64), the rest takes the literal chain
    const size_t cap3 = std::max<size_t>(WAVE, cap / 16);
    const size_t list_bytes = ((size_t)3 * FB_SUB * cap + (size_t)FB_SUB * cap3) * sizeof(uint32_t);
    const size_t cnt_bytes = (size_t)FB_KINDS * FB_SUB * FB_PAD * sizeof(int);
    const size_t b_off = (list_bytes + cnt_bytes + 255) & ~(size_t)255;
    WM_TRY(grow(ctx, &ctx->fb_list, &ctx->fb_bytes, b_off + (size_t)FB_SUB * cap3 * 64 * sizeof(float), "fallback lists"));
    uint32_t* fb = (uint32_t*)ctx->fb_list;
    int* fb_cnt = (int*)((char*)ctx->fb_list + list_bytes);
    float* fb_b = (float*)((char*)ctx->fb_list + b_off);
    const uint32_t fb_cap3 = (uint32_t)cap3;
    WM_HIP(hipMemsetAsync(fb_cnt, 0, cnt_bytes, ctx->stream));
    const uint32_t fb_cap = (uint32_t)cap;
    const dim3 fgrid((unsigned)(n_waves < 2048 ? n_waves : 2048));
#define WM_LAUNCH_EMBED(A, Y)                                                                      \
  do {                                                                                             \
    hipLaunchKernelGGL((k_embed_tiles<A, Y>), grid, block, 0, ctx->stream, host, sigma_w, stego,   \
                       sigma_c, yw, g, n_groups, sigma_w_plane_stride, alpha, K,                   \
                       ctx->d_status, fb, fb_cnt, fb_cap, fb_b, fb_cap3);                           \
    hipLaunchKernelGGL((k_embed_fallback<A, Y>), fgrid, block, 0, ctx->stream, host, sigma_w,      \
                       stego, sigma_c, yw, g, sigma_w_plane_stride, alpha, K, ctx->d_status, fb,   \
                       fb_cnt, fb_cap, fb_b, fb_cap3);                                             \
    hipLaunchKernelGGL((k_embed_one_small<A, Y>), fgrid, block, 0, ctx->stream, host, sigma_w,     \
                       stego, sigma_c, yw, g, sigma_w_plane_stride, alpha, K, ctx->d_status, fb,   \
                       fb_cnt, fb_cap, fb_b, fb_cap3);                                             \
  } while (0)
    if (al && yw) WM_LAUNCH_EMBED(true, true);
    else if (al) WM_LAUNCH_EMBED(true, false);
    else if (yw) WM_LAUNCH_EMBED(false, true);
    else WM_LAUNCH_EMBED(false, false);
#undef WM_LAUNCH_EMBED
    WM_HIP(hipGetLastError());
  }
  const int Hb = (H / 8) * 8, Wb = (W / 8) * 8;
  if ((Hb != H || Wb != W) && (stego != host || yw)) {
    const int n = (W - Wb) * Hb + (H - Hb) * W;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)n_planes);
    hipLaunchKernelGGL(k_copy_border, grid, dim3(256), 0, ctx->stream, host, stego, yw, H, W, Hb, Wb,
                       (size_t)row_stride, plane_stride);
    WM_HIP(hipGetLastError());
  }
  return WM_OK;
}

// ---- K2 --------------------------------------------------------------------
int wm_sigma_tiles_u8_dev(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H,
                          int W, int row_stride, size_t plane_stride) {
  WM_TRY(check_plane_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  if (n_planes == 0 || g.n_tiles == 0) return WM_OK;
  if (!sigma || ((uintptr_t)sigma & 15u)) return set_err(WM_ERR_BADARG, "sigma is NULL or not 16-byte aligned");
  const dim3 grid = tile_grid(g, n_planes), block(WAVE);
  if (u8_aligned(planes, planes, row_stride, plane_stride))
    hipLaunchKernelGGL((k_sigma_tiles<true>), grid, block, 0, ctx->stream, planes, sigma, g, ctx->d_status);
  else
    hipLaunchKernelGGL((k_sigma_tiles<false>), grid, block, 0, ctx->stream, planes, sigma, g, ctx->d_status);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ---- K3 --------------------------------------------------------------------
int wm_svd_tiles_f32_dev(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt,
                         int n_planes, int H, int W, int row_stride, size_t plane_stride) {
  WM_TRY(check_plane_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  if (n_planes == 0 || g.n_tiles == 0) return WM_OK;
  if (!U || !S || !Vt || (((uintptr_t)U | (uintptr_t)S | (uintptr_t)Vt) & 15u))
    return set_err(WM_ERR_BADARG, "U/S/Vt is NULL or not 16-byte aligned");
  const dim3 grid = tile_grid(g, n_planes), block(WAVE);
  if (f32_vec_ok(planes, (size_t)row_stride, plane_stride))
    hipLaunchKernelGGL((k_svd_tiles<true>), grid, block, 0, ctx->stream, planes, U, S, Vt, g, ctx->d_status);
  else
    hipLaunchKernelGGL((k_svd_tiles<false>), grid, block, 0, ctx->stream, planes, U, S, Vt, g, ctx->d_status);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ---- K2+K4 -----------------------------------------------------------------
static int extract_tiles_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                             const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                             size_t plane_stride, size_t uv_plane_stride, float alpha, int K, bool px, unsigned* mm = nullptr) {
  WM_TRY(check_plane_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!out) return set_err(WM_ERR_BADARG, "out is NULL");
  if (K < 0 || K > 8) return set_err(WM_ERR_BADARG, "K must be in 0..8");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  if ((H % 8) || (W % 8))
    WM_HIP(hipMemsetAsync(out, 0, (size_t)n_planes * g.HW * sizeof(float), ctx->stream));
  if (g.n_tiles == 0) return WM_OK;
  if (!sigma_c || !Uw || !Vwt || (((uintptr_t)sigma_c | (uintptr_t)Uw | (uintptr_t)Vwt) & 15u))
    return set_err(WM_ERR_BADARG, "sigma_c/Uw/Vwt is NULL or not 16-byte aligned");
  if (uv_plane_stride != 0 && uv_plane_stride != (size_t)g.n_tiles)
    return set_err(WM_ERR_BADARG, "uv_plane_stride must be 0 (shared) or n_tiles");
  const float inv_alpha = 1.0f / fmaxf(alpha, 1e-8f);
  const bool al = u8_aligned(stego, stego, row_stride, plane_stride);
  const bool vf = f32_vec_ok(out, (size_t)W, g.HW);
  if ((((size_t)g.n_tiles + WAVE - 1) / WAVE + N_XCD) * (size_t)n_planes > 0x7fffffffull)
    return set_err(WM_ERR_BADARG, "more than 2^31 tile groups in one call");
  const dim3 grid = tile_grid_planefast(g, n_planes), block(WAVE);
#define WM_LAUNCH_EXTRACT(A, V, P)                                                                    \
  do {                                                                                                \
    if (mm) hipLaunchKernelGGL((k_extract_tiles<A, V, P, true>), grid, block, 0, ctx->stream, stego, sigma_c, Uw, Vwt, \
                               out, g, (unsigned)n_planes, uv_plane_stride, inv_alpha, K, ctx->d_status, mm); \
    else hipLaunchKernelGGL((k_extract_tiles<A, V, P, false>), grid, block, 0, ctx->stream, stego, sigma_c, Uw, Vwt, \
                            out, g, (unsigned)n_planes, uv_plane_stride, inv_alpha, K, ctx->d_status, mm); \
  } while (0)
  if (px) {
    if (al && vf) WM_LAUNCH_EXTRACT(true, true, true);
    else if (al) WM_LAUNCH_EXTRACT(true, false, true);
    else if (vf) WM_LAUNCH_EXTRACT(false, true, true);
    else WM_LAUNCH_EXTRACT(false, false, true);
  } else {
    if (al && vf) WM_LAUNCH_EXTRACT(true, true, false);
    else if (al) WM_LAUNCH_EXTRACT(true, false, false);
    else if (vf) WM_LAUNCH_EXTRACT(false, true, false);
    else WM_LAUNCH_EXTRACT(false, false, false);
  }
#undef WM_LAUNCH_EXTRACT
  WM_HIP(hipGetLastError());
  return WM_OK;
}

int wm_extract_tiles_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                            const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                            size_t plane_stride, size_t uv_plane_stride, float alpha, int K) {
  return extract_tiles_dev(ctx, stego, sigma_c, Uw, Vwt, out, n_planes, H, W, row_stride, plane_stride,
                           uv_plane_stride, alpha, K, false);
}

int wm_extract_tiles_px_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Ux,
                               const float* Vxt, float* out, int n_planes, int H, int W, int row_stride,
                               size_t plane_stride, size_t uv_plane_stride, float alpha, int K) {
  return extract_tiles_dev(ctx, stego, sigma_c, Ux, Vxt, out, n_planes, H, W, row_stride, plane_stride,
                           uv_plane_stride, alpha, K, true);
}

// single:203-222 per plane in one call: sigma + rank-8 product (K2+K4) -> routed unscramble -> min-max normalise -> uint8.
// The float estimate lives in a grow-only buffer of the context; the min / max come out of the extract kernel itself.
int wm_extract_unscrambled_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw, const float* Vwt,
                                  const wm_route* route, uint8_t* out, int n_planes, int H, int W, int row_stride,
                                  size_t plane_stride, size_t uv_plane_stride, float alpha, int K, int px, int do_norm) {
  WM_TRY(check_plane_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!out || !route) return set_err(WM_ERR_BADARG, "out / route is NULL");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const size_t n = (size_t)H * W;
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  const size_t groups = ((size_t)g.n_tiles + WAVE - 1) / WAVE;
  const size_t f_bytes = ((size_t)n_planes * n * sizeof(float) + 255) & ~(size_t)255;
  WM_TRY(grow(ctx, &ctx->extract_f32, &ctx->extract_f32_bytes, f_bytes + (size_t)n_planes * (groups + 1) * 2 * sizeof(unsigned) + 256,
              "extract staging"));
  float* w = (float*)ctx->extract_f32;
  unsigned* mm = (unsigned*)((char*)ctx->extract_f32 + f_bytes);
  const bool use_mm = do_norm && g.n_tiles > 0;
  WM_TRY(extract_tiles_dev(ctx, stego, sigma_c, Uw, Vwt, w, n_planes, H, W, row_stride, plane_stride, uv_plane_stride, alpha, K,
                           px != 0, use_mm ? mm : nullptr));
  // pixels outside the tile grid (H % 8, W % 8) are zeros of the estimate: they take part in the min / max
  return wmi::route_unpermute_normalize(ctx, w, route, out, n, n_planes, do_norm, use_mm ? mm : nullptr, (unsigned)groups,
                                        ((H % 8) || (W % 8)) ? 1 : 0);
}

int wm_tile_factors_to_pixel_dev(wm_ctx* ctx, const float* Uw, const float* Vwt, float* Ux, float* Vxt,
                                 size_t n_tiles) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_tiles == 0) return WM_OK;
  if (!Uw || !Vwt || !Ux || !Vxt || (((uintptr_t)Uw | (uintptr_t)Vwt | (uintptr_t)Ux | (uintptr_t)Vxt) & 15u))
    return set_err(WM_ERR_BADARG, "factor arrays are NULL or not 16-byte aligned");
  if (n_tiles > ((size_t)1 << 31)) return set_err(WM_ERR_BADARG, "too many tiles");
  hipLaunchKernelGGL(k_factors_to_pixel, dim3((unsigned)((n_tiles + WAVE - 1) / WAVE)), dim3(WAVE), 0, ctx->stream,
                     Uw, Vwt, Ux, Vxt, n_tiles);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ---- K4 --------------------------------------------------------------------
int wm_reconstruct_tiles_dev(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt,
                             float* out, int n_planes, int H, int W) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || H < 0 || W < 0 || n_planes > 65535) return set_err(WM_ERR_BADARG, "bad size");
  if (!out) return set_err(WM_ERR_BADARG, "out is NULL");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const Geom g = make_geom(H, W, W, (size_t)H * W);
  if ((H % 8) || (W % 8))
    WM_HIP(hipMemsetAsync(out, 0, (size_t)n_planes * g.HW * sizeof(float), ctx->stream));
  if (g.n_tiles == 0) return WM_OK;
  if (!Uw || !sw_hat || !Vwt || (((uintptr_t)Uw | (uintptr_t)sw_hat | (uintptr_t)Vwt) & 15u))
    return set_err(WM_ERR_BADARG, "Uw/sw_hat/Vwt is NULL or not 16-byte aligned");
  const dim3 grid = tile_grid(g, n_planes), block(WAVE);
  if (f32_vec_ok(out, (size_t)W, g.HW))
    hipLaunchKernelGGL((k_reconstruct_tiles<true>), grid, block, 0, ctx->stream, Uw, sw_hat, Vwt, out, g);
  else
    hipLaunchKernelGGL((k_reconstruct_tiles<false>), grid, block, 0, ctx->stream, Uw, sw_hat, Vwt, out, g);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ---- K2+K5 -----------------------------------------------------------------
int wm_detect_tiles_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c,
                           const float* sigma_w, double* scores, int n_planes, int H, int W,
                           int row_stride, size_t plane_stride, size_t sigma_w_plane_stride,
                           float alpha) {
  WM_TRY(check_plane_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!scores) return set_err(WM_ERR_BADARG, "scores is NULL");
  if (n_planes == 0) return WM_OK;
  const Geom g = make_geom(H, W, row_stride, plane_stride);
  const int n_waves = (g.n_tiles + WAVE - 1) / WAVE;
  if (g.n_tiles > 0) {
    if (!sigma_c || !sigma_w || (((uintptr_t)sigma_c | (uintptr_t)sigma_w) & 15u) ||
        (sigma_w_plane_stride & 3u))
      return set_err(WM_ERR_BADARG, "sigma_c/sigma_w is NULL or not 16-byte aligned");
    WM_TRY(grow(ctx, &ctx->partials, &ctx->partials_bytes,
                (size_t)n_planes * n_waves * N_SUMS * sizeof(double), "detect partial sums"));
    const float inv_alpha = 1.0f / fmaxf(alpha, 1e-8f);
    const dim3 grid = tile_grid(g, n_planes), block(WAVE);
    if (u8_aligned(stego, stego, row_stride, plane_stride))
      hipLaunchKernelGGL((k_detect_tiles<true>), grid, block, 0, ctx->stream, stego, sigma_c, sigma_w,
                         (double*)ctx->partials, g, sigma_w_plane_stride, inv_alpha, ctx->d_status);
    else
      hipLaunchKernelGGL((k_detect_tiles<false>), grid, block, 0, ctx->stream, stego, sigma_c, sigma_w,
                         (double*)ctx->partials, g, sigma_w_plane_stride, inv_alpha, ctx->d_status);
    WM_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_detect_finalize, dim3((unsigned)n_planes), dim3(256), 0, ctx->stream,
                     (const double*)ctx->partials, scores, n_waves, (double)g.n_tiles * 8.0);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ===========================================================================
// host-pointer wrappers: H2D, kernels, D2H, sync.  Convenience for callers
// without their own device allocator; the throughput path is *_dev.
// ===========================================================================

int wm_embed_tiles_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego,
                      float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                      size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K) {
  WM_TRY(check_plane_args(ctx, host, n_planes, H, W, row_stride, plane_stride));
  if (!stego) return set_err(WM_ERR_BADARG, "stego is NULL");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (nt > 0 && (!sigma_w || !sigma_c)) return set_err(WM_ERR_BADARG, "sigma_w / sigma_c is NULL");
  const size_t span = plane_span(n_planes, H, row_stride, plane_stride, W);
  const size_t n_sw = sigma_w_plane_stride ? (size_t)(n_planes - 1) * sigma_w_plane_stride + nt * 8 : nt * 8;
  const size_t n_sc = (size_t)n_planes * nt * 8;
  const size_t n_yw = yw ? (size_t)n_planes * H * W : 0;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes,
              2 * pad256(span) + pad256(n_sw * 4) + pad256(n_sc * 4) + pad256(n_yw * 4) + 2048, "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  uint8_t* d_host = cv.take<uint8_t>(span);
  uint8_t* d_stego = cv.take<uint8_t>(span);
  float* d_sw = cv.take<float>(n_sw);
  float* d_sc = cv.take<float>(n_sc);
  float* d_yw = yw ? cv.take<float>(n_yw) : nullptr;
  WM_HIP(hipMemcpyAsync(d_host, host, span, hipMemcpyHostToDevice, ctx->stream));
  // bytes between rows / planes that belong to the caller must survive the round trip
  if (stego != host) WM_HIP(hipMemcpyAsync(d_stego, stego, span, hipMemcpyHostToDevice, ctx->stream));
  else d_stego = d_host;
  if (nt) WM_HIP(hipMemcpyAsync(d_sw, sigma_w, n_sw * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_embed_tiles_u8_dev(ctx, d_host, d_sw, d_stego, d_sc, d_yw, n_planes, H, W, row_stride,
                               plane_stride, sigma_w_plane_stride, alpha, K));
  WM_HIP(hipMemcpyAsync(stego, d_stego, span, hipMemcpyDeviceToHost, ctx->stream));
  if (nt) WM_HIP(hipMemcpyAsync(sigma_c, d_sc, n_sc * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (yw) WM_HIP(hipMemcpyAsync(yw, d_yw, n_yw * 4, hipMemcpyDeviceToHost, ctx->stream));
  return wm_check_status(ctx);
}

int wm_sigma_tiles_u8(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H, int W,
                      int row_stride, size_t plane_stride) {
  WM_TRY(check_plane_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (n_planes == 0 || nt == 0) return WM_OK;
  if (!sigma) return set_err(WM_ERR_BADARG, "sigma is NULL");
  const size_t span = plane_span(n_planes, H, row_stride, plane_stride, W);
  const size_t n_s = (size_t)n_planes * nt * 8;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes, pad256(span) + pad256(n_s * 4) + 1024, "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  uint8_t* d_p = cv.take<uint8_t>(span);
  float* d_s = cv.take<float>(n_s);
  WM_HIP(hipMemcpyAsync(d_p, planes, span, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_sigma_tiles_u8_dev(ctx, d_p, d_s, n_planes, H, W, row_stride, plane_stride));
  WM_HIP(hipMemcpyAsync(sigma, d_s, n_s * 4, hipMemcpyDeviceToHost, ctx->stream));
  return wm_check_status(ctx);
}

int wm_svd_tiles_f32(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt, int n_planes,
                     int H, int W, int row_stride, size_t plane_stride) {
  WM_TRY(check_plane_args(ctx, planes, n_planes, H, W, row_stride, plane_stride));
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (n_planes == 0 || nt == 0) return WM_OK;
  if (!U || !S || !Vt) return set_err(WM_ERR_BADARG, "U/S/Vt is NULL");
  const size_t span = plane_span(n_planes, H, row_stride, plane_stride, W);
  const size_t n_m = (size_t)n_planes * nt * 64, n_s = (size_t)n_planes * nt * 8;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes,
              pad256(span * 4) + 2 * pad256(n_m * 4) + pad256(n_s * 4) + 2048, "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  float* d_p = cv.take<float>(span);
  float* d_U = cv.take<float>(n_m);
  float* d_V = cv.take<float>(n_m);
  float* d_S = cv.take<float>(n_s);
  WM_HIP(hipMemcpyAsync(d_p, planes, span * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_svd_tiles_f32_dev(ctx, d_p, d_U, d_S, d_V, n_planes, H, W, row_stride, plane_stride));
  WM_HIP(hipMemcpyAsync(U, d_U, n_m * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipMemcpyAsync(Vt, d_V, n_m * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipMemcpyAsync(S, d_S, n_s * 4, hipMemcpyDeviceToHost, ctx->stream));
  return wm_check_status(ctx);
}

static int extract_tiles_host(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                              const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                              size_t plane_stride, size_t uv_plane_stride, float alpha, int K, bool sum_planes) {
  WM_TRY(check_plane_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!out) return set_err(WM_ERR_BADARG, "out is NULL");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (nt > 0 && (!sigma_c || !Uw || !Vwt)) return set_err(WM_ERR_BADARG, "sigma_c/Uw/Vwt is NULL");
  const size_t span = plane_span(n_planes, H, row_stride, plane_stride, W);
  const size_t n_sc = (size_t)n_planes * nt * 8;
  const size_t n_uv = (uv_plane_stride ? (size_t)n_planes : (size_t)1) * nt * 64;
  const size_t hw = (size_t)H * W, n_out = (size_t)n_planes * hw;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes,
              pad256(span) + pad256(n_sc * 4) + 2 * pad256(n_uv * 4) + pad256(n_out * 4) + pad256(hw * 4) + 2048, "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  uint8_t* d_p = cv.take<uint8_t>(span);
  float* d_sc = cv.take<float>(n_sc);
  float* d_U = cv.take<float>(n_uv);
  float* d_V = cv.take<float>(n_uv);
  float* d_o = cv.take<float>(n_out);
  float* d_sum = cv.take<float>(hw);
  WM_HIP(hipMemcpyAsync(d_p, stego, span, hipMemcpyHostToDevice, ctx->stream));
  if (nt) {
    WM_HIP(hipMemcpyAsync(d_sc, sigma_c, n_sc * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(d_U, Uw, n_uv * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(d_V, Vwt, n_uv * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  WM_TRY(wm_extract_tiles_u8_dev(ctx, d_p, d_sc, d_U, d_V, d_o, n_planes, H, W, row_stride, plane_stride,
                                 uv_plane_stride, alpha, K));
  if (sum_planes) {
    const size_t g = (hw + 255) / 256;
    hipLaunchKernelGGL(k_sum_planes, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, ctx->stream, d_o, hw, n_planes, d_sum);
    WM_HIP(hipGetLastError());
    WM_HIP(hipMemcpyAsync(out, d_sum, hw * 4, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    WM_HIP(hipMemcpyAsync(out, d_o, n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  return wm_check_status(ctx);
}

int wm_extract_tiles_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                        const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                        size_t plane_stride, size_t uv_plane_stride, float alpha, int K) {
  return extract_tiles_host(ctx, stego, sigma_c, Uw, Vwt, out, n_planes, H, W, row_stride, plane_stride,
                            uv_plane_stride, alpha, K, false);
}

int wm_extract_tiles_sum_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                            const float* Vwt, float* out_sum, int n_planes, int H, int W, int row_stride,
                            size_t plane_stride, size_t uv_plane_stride, float alpha, int K) {
  return extract_tiles_host(ctx, stego, sigma_c, Uw, Vwt, out_sum, n_planes, H, W, row_stride, plane_stride,
                            uv_plane_stride, alpha, K, true);
}

int wm_reconstruct_tiles(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt,
                         float* out, int n_planes, int H, int W) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || H < 0 || W < 0) return set_err(WM_ERR_BADARG, "negative size");
  if (!out) return set_err(WM_ERR_BADARG, "out is NULL");
  if (n_planes == 0 || H == 0 || W == 0) return WM_OK;
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (nt > 0 && (!Uw || !sw_hat || !Vwt)) return set_err(WM_ERR_BADARG, "Uw/sw_hat/Vwt is NULL");
  const size_t n_m = (size_t)n_planes * nt * 64, n_s = (size_t)n_planes * nt * 8;
  const size_t n_out = (size_t)n_planes * H * W;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes,
              2 * pad256(n_m * 4) + pad256(n_s * 4) + pad256(n_out * 4) + 2048, "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  float* d_U = cv.take<float>(n_m);
  float* d_V = cv.take<float>(n_m);
  float* d_s = cv.take<float>(n_s);
  float* d_o = cv.take<float>(n_out);
  if (nt) {
    WM_HIP(hipMemcpyAsync(d_U, Uw, n_m * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(d_V, Vwt, n_m * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(d_s, sw_hat, n_s * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  WM_TRY(wm_reconstruct_tiles_dev(ctx, d_U, d_s, d_V, d_o, n_planes, H, W));
  WM_HIP(hipMemcpyAsync(out, d_o, n_out * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_detect_tiles_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                       double* scores, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                       size_t sigma_w_plane_stride, float alpha) {
  WM_TRY(check_plane_args(ctx, stego, n_planes, H, W, row_stride, plane_stride));
  if (!scores) return set_err(WM_ERR_BADARG, "scores is NULL");
  if (n_planes == 0) return WM_OK;
  const size_t nt = (size_t)(H / 8) * (W / 8);
  if (nt > 0 && (!sigma_c || !sigma_w)) return set_err(WM_ERR_BADARG, "sigma_c/sigma_w is NULL");
  const size_t span = plane_span(n_planes, H, row_stride, plane_stride, W);
  const size_t n_sc = (size_t)n_planes * nt * 8;
  const size_t n_sw = sigma_w_plane_stride ? (size_t)(n_planes - 1) * sigma_w_plane_stride + nt * 8 : nt * 8;
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes,
              pad256(span) + pad256(n_sc * 4) + pad256(n_sw * 4) + pad256((size_t)n_planes * 8) + 2048,
              "scratch"));
  Carve cv{(char*)ctx->scratch, 0};
  uint8_t* d_p = cv.take<uint8_t>(span ? span : 1);
  float* d_sc = cv.take<float>(n_sc);
  float* d_sw = cv.take<float>(n_sw);
  double* d_scores = cv.take<double>((size_t)n_planes);
  if (span) WM_HIP(hipMemcpyAsync(d_p, stego, span, hipMemcpyHostToDevice, ctx->stream));
  if (nt) {
    WM_HIP(hipMemcpyAsync(d_sc, sigma_c, n_sc * 4, hipMemcpyHostToDevice, ctx->stream));
    WM_HIP(hipMemcpyAsync(d_sw, sigma_w, n_sw * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  WM_TRY(wm_detect_tiles_u8_dev(ctx, d_p, d_sc, d_sw, d_scores, n_planes, H, W, row_stride, plane_stride,
                                sigma_w_plane_stride, alpha));
  WM_HIP(hipMemcpyAsync(scores, d_scores, (size_t)n_planes * 8, hipMemcpyDeviceToHost, ctx->stream));
  return wm_check_status(ctx);
}

}  // extern "C"
