// Pixel-side kernels either side of the hot path:
// colour conversion in OpenCV's 8-bit fixed point (bit-exact integer work),
// PSNR, SSIM (11x11 sigma-1.5 Gaussian stencil, reflect-101 border) and
// min-max normalisation.  All of them are HBM-bound streaming kernels.
//
// Reference statements (app_dct_svd_single.py): _to_Y / _from_Y cvtColor
// (single:21-30), psnr (single:38-42), ssim (single:44-57), cv2.normalize
// NORM_MINMAX (single:221, 269-271), BGR2GRAY (single:45, 170, 190).
#include <math.h>

#include "wm_internal.h"

using namespace wmi;

namespace {

// ---- colour: 16 pixels (48 bytes = 3 x 16-byte words) per thread ----------------
__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)min(max(v, 0), 255); }
// (v >> 14) saturated to 0..255.  The empty asm keeps hipcc (ROCm 7.2) from fusing
// shift + clamp pairs into v_ashr_pk_u8_i32: it then ORs further bytes into the
// same register assuming bits 16..31 of that instruction's result are zero, which
// they are not on gfx950 (observed: byte 2 of every 4th output word corrupted).
__device__ __forceinline__ uint32_t shr14_sat_u8(int v) {
  int s = v >> 14;
  asm volatile("" : "+v"(s));
  return clamp255(s);
}

// cv2.COLOR_BGR2YCrCb, 8-bit: yuv_shift 14, coefficients 4899 / 9617 / 1868, 11682, 9241
__device__ __forceinline__ void bgr2ycc(const uint32_t b, const uint32_t g, const uint32_t r, uint32_t& y,
                                        uint32_t& cr, uint32_t& cb) {
  const int Y = (int)(r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14;   // 0..255 by construction
  cr = shr14_sat_u8(((int)r - Y) * 11682 + (128 << 14) + 8192);
  cb = shr14_sat_u8(((int)b - Y) * 9241 + (128 << 14) + 8192);
  y = (uint32_t)Y;
}
// cv2.COLOR_YCrCb2BGR, 8-bit: 22987, -11698, -5636, 29049
__device__ __forceinline__ void ycc2bgr(const uint32_t y, const uint32_t cr, const uint32_t cb, uint32_t& b,
                                        uint32_t& g, uint32_t& r) {
  const int c_r = (int)cr - 128, c_b = (int)cb - 128;
  // y + (x >> 14) == ((y << 14) + x) >> 14 for an arithmetic shift
  b = shr14_sat_u8(((int)y << 14) + c_b * 29049 + 8192);
  g = shr14_sat_u8(((int)y << 14) + c_b * -5636 + c_r * -11698 + 8192);
  r = shr14_sat_u8(((int)y << 14) + c_r * 22987 + 8192);
}
// cv2.COLOR_BGR2GRAY, 8-bit: (B*3735 + G*19235 + R*9798 + 2^14) >> 15
__device__ __forceinline__ uint32_t bgr2gray(const uint32_t b, const uint32_t g, const uint32_t r) {
  return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
}

enum ColorOp { BGR_TO_YCC = 0, YCC_TO_BGR = 1, BGR_TO_GRAY = 2, BGR_TO_Y = 3, Y_INTO_YCC_TO_BGR = 4 };

__device__ __forceinline__ uint32_t byte_of(const uint32_t (&w)[12], const int i) {
  return (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
}

// one group of 16 pixels: w = its 48 interleaved input bytes, pl = 16 plane bytes (replace-Y only) -> o (48 interleaved
// output bytes) or po (16 plane bytes)
template <int OP>
__device__ __forceinline__ void color_group(const uint32_t (&w)[12], const uint32_t (&pl)[4], uint32_t (&o)[12], uint32_t (&po)[4]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) o[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) po[i] = 0;
#pragma unroll
  for (int px = 0; px < 16; ++px) {
    const uint32_t c0 = byte_of(w, 3 * px), c1 = byte_of(w, 3 * px + 1), c2 = byte_of(w, 3 * px + 2);
    uint32_t a0 = 0, a1 = 0, a2 = 0, p1 = 0;
    if (OP == BGR_TO_YCC) bgr2ycc(c0, c1, c2, a0, a1, a2);
    else if (OP == YCC_TO_BGR) ycc2bgr(c0, c1, c2, a0, a1, a2);
    else if (OP == BGR_TO_GRAY) p1 = bgr2gray(c0, c1, c2);
    else if (OP == BGR_TO_Y) { uint32_t cr, cb; bgr2ycc(c0, c1, c2, p1, cr, cb); }
    else {  // replace Y of the BGR pixel's YCrCb by the plane value, convert back (single:26-30)
      uint32_t y, cr, cb; bgr2ycc(c0, c1, c2, y, cr, cb);
      const uint32_t ynew = (pl[px >> 2] >> (8 * (px & 3))) & 0xffu;
      ycc2bgr(ynew, cr, cb, a0, a1, a2);
    }
    if (OP == BGR_TO_GRAY || OP == BGR_TO_Y) po[px >> 2] |= p1 << (8 * (px & 3));
    else {
      o[(3 * px) >> 2] |= a0 << (8 * ((3 * px) & 3));
      o[(3 * px + 1) >> 2] |= a1 << (8 * ((3 * px + 1) & 3));
      o[(3 * px + 2) >> 2] |= a2 << (8 * ((3 * px + 2) & 3));
    }
  }
}

// generic over pixel groups: in3 = interleaved 3-channel input (16 px = 48 B), plane = single-channel
// input/output (16 B per group), out3 = interleaved 3-channel output
template <int OP>
__global__ __launch_bounds__(256) void k_color(const uint8_t* __restrict__ in3, const uint8_t* __restrict__ plane_in,
                                              uint8_t* __restrict__ out3, uint8_t* __restrict__ plane_out,
                                              const size_t n_px) {
  constexpr bool OUT3 = !(OP == BGR_TO_GRAY || OP == BGR_TO_Y);
  const size_t n_groups = n_px / 16;
  auto load_group = [&](const size_t gidx, uint32_t (&w)[12], uint32_t (&pl)[4]) {
    // the thread's own 48 bytes (a line is shared by three load instructions of the wave; nothing is fetched twice)
    const uint4* src = reinterpret_cast<const uint4*>(in3 + gidx * 48);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const uint4 v = src[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    pl[0] = pl[1] = pl[2] = pl[3] = 0;
    if (OP == Y_INTO_YCC_TO_BGR) {
      const uint4 v = *reinterpret_cast<const uint4*>(plane_in + gidx * 16);
      pl[0] = v.x; pl[1] = v.y; pl[2] = v.z; pl[3] = v.w;
    }
  };
  // Interleaved 3-channel OUTPUT leaves the workgroup as whole 16-byte chunks in address order (a wave's store instruction
  // covers 1 KiB contiguous): the thread that owns a 16-pixel group puts its 48 bytes - chunks 3t .. 3t + 2 of the block's
  // 12 KiB - into LDS (ds_write_b128 at a 48-byte lane stride: conflict-free) and the block stores the chunks in order.
  // Direct stores of the own 48 bytes touch 24 lines per wave instruction for 1 KiB of data: BGR -> YCrCb 59 -> 67 % of
  // 8 TB/s, replace-Y 63 -> 66 % (profiles/r03y_color_lds.log).  The same staging on the LOAD side changes nothing or costs
  // (BGR -> Y 77 -> 76 %): the loads stay direct, and so do the 16-byte plane stores.
  __shared__ uint4 stage[OUT3 ? 3 * 256 : 1];
  const size_t n_full = OUT3 ? n_groups / 256 : 0;       // whole blocks of 256 groups; the rest takes the direct form below
  for (size_t blk = blockIdx.x; blk < n_full; blk += gridDim.x) {
    uint32_t w[12], pl[4], o[12], po[4];
    load_group(blk * 256 + threadIdx.x, w, pl);
    color_group<OP>(w, pl, o, po);
#pragma unroll
    for (int i = 0; i < 3; ++i) stage[3 * threadIdx.x + i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(out3 + blk * 256 * 48);
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[i * 256 + threadIdx.x] = stage[i * 256 + threadIdx.x];
    __syncthreads();                                     // the next block's groups overwrite stage
  }
  // groups outside whole blocks, and every group of the single-plane outputs: the direct form (kept literally as it was -
  // through the shared helpers the BGR -> Y loop came out at 44 instead of 61 VGPRs and 71 instead of 78 % of 8 TB/s)
  for (size_t gidx = n_full * 256 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < n_groups;
       gidx += (size_t)gridDim.x * blockDim.x) {
    uint32_t w[12];
    const uint4* src = reinterpret_cast<const uint4*>(in3 + gidx * 48);
#pragma unroll
    for (int i = 0; i < 3; ++i) { const uint4 v = src[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    uint32_t pl[4] = {0, 0, 0, 0};
    if (OP == Y_INTO_YCC_TO_BGR) {
      const uint4 v = *reinterpret_cast<const uint4*>(plane_in + gidx * 16);
      pl[0] = v.x; pl[1] = v.y; pl[2] = v.z; pl[3] = v.w;
    }
    uint32_t o[12] = {};
    uint32_t po[4] = {0, 0, 0, 0};
#pragma unroll
    for (int px = 0; px < 16; ++px) {
      const uint32_t c0 = byte_of(w, 3 * px), c1 = byte_of(w, 3 * px + 1), c2 = byte_of(w, 3 * px + 2);
      uint32_t a0 = 0, a1 = 0, a2 = 0, p1 = 0;
      if (OP == BGR_TO_YCC) bgr2ycc(c0, c1, c2, a0, a1, a2);
      else if (OP == YCC_TO_BGR) ycc2bgr(c0, c1, c2, a0, a1, a2);
      else if (OP == BGR_TO_GRAY) p1 = bgr2gray(c0, c1, c2);
      else if (OP == BGR_TO_Y) { uint32_t cr, cb; bgr2ycc(c0, c1, c2, p1, cr, cb); }
      else {  // replace Y of the BGR pixel's YCrCb by the plane value, convert back (single:26-30)
        uint32_t y, cr, cb; bgr2ycc(c0, c1, c2, y, cr, cb);
        const uint32_t ynew = (pl[px >> 2] >> (8 * (px & 3))) & 0xffu;
        ycc2bgr(ynew, cr, cb, a0, a1, a2);
      }
      if (OP == BGR_TO_GRAY || OP == BGR_TO_Y) po[px >> 2] |= p1 << (8 * (px & 3));
      else {
        o[(3 * px) >> 2] |= a0 << (8 * ((3 * px) & 3));
        o[(3 * px + 1) >> 2] |= a1 << (8 * ((3 * px + 1) & 3));
        o[(3 * px + 2) >> 2] |= a2 << (8 * ((3 * px + 2) & 3));
      }
    }
    if (OP == BGR_TO_GRAY || OP == BGR_TO_Y) {
      *reinterpret_cast<uint4*>(plane_out + gidx * 16) = make_uint4(po[0], po[1], po[2], po[3]);
    } else {
      uint4* dst = reinterpret_cast<uint4*>(out3 + gidx * 48);
#pragma unroll
      for (int i = 0; i < 3; ++i) dst[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
  }
  // tail (n_px % 16 pixels): one thread, scalar
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (size_t px = n_groups * 16; px < n_px; ++px) {
      const uint32_t c0 = in3[px * 3], c1 = in3[px * 3 + 1], c2 = in3[px * 3 + 2];
      uint32_t a0 = 0, a1 = 0, a2 = 0, p1 = 0;
      if (OP == BGR_TO_YCC) bgr2ycc(c0, c1, c2, a0, a1, a2);
      else if (OP == YCC_TO_BGR) ycc2bgr(c0, c1, c2, a0, a1, a2);
      else if (OP == BGR_TO_GRAY) p1 = bgr2gray(c0, c1, c2);
      else if (OP == BGR_TO_Y) { uint32_t cr, cb; bgr2ycc(c0, c1, c2, p1, cr, cb); }
      else { uint32_t y, cr, cb; bgr2ycc(c0, c1, c2, y, cr, cb); ycc2bgr(plane_in[px], cr, cb, a0, a1, a2); }
      if (OP == BGR_TO_GRAY || OP == BGR_TO_Y) plane_out[px] = (uint8_t)p1;
      else { out3[px * 3] = (uint8_t)a0; out3[px * 3 + 1] = (uint8_t)a1; out3[px * 3 + 2] = (uint8_t)a2; }
    }
  }
}

// ---- PSNR: sum of squared differences of two uint8 buffers ------------------------
__global__ __launch_bounds__(256) void k_sqdiff_u8(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                  const size_t n, unsigned long long* __restrict__ out) {
  // sum (x - y)^2 = sum x^2 + sum y^2 - 2 sum xy: three v_dot4_u32_u8 per 4 byte pairs instead of 4 x (two field
  // extracts, subtract, multiply, add) - the byte-wise form was VALU-bound at 2.6 TB/s (profiles/r03_p_pixel_summary.json)
  unsigned long long acc = 0;
  const size_t n16 = n / 16;
  const uint4* a4 = reinterpret_cast<const uint4*>(a);
  const uint4* b4 = reinterpret_cast<const uint4*>(b);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = a4[i], y = b4[i];
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
    uint32_t sq = 0, xy = 0;           // at most 32 x 255^2 and 16 x 255^2 per iteration
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sq = __builtin_amdgcn_udot4(xs[w], xs[w], sq, false);
      sq = __builtin_amdgcn_udot4(ys[w], ys[w], sq, false);
      xy = __builtin_amdgcn_udot4(xs[w], ys[w], xy, false);
    }
    acc += sq - 2u * xy;               // >= 0: it is the sum of 16 squares
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n16 * 16; i < n; ++i) { const int d = (int)a[i] - (int)b[i]; acc += (unsigned)(d * d); }
  // one partial per workgroup, summed by k_sum_u64: 8 192 same-address atomics from 8 XCDs serialised at ~18 ns each and
  // WERE the kernel's 150 us (2.6 TB/s whatever the arithmetic or the operands' relative alignment - tools/sqdiff_offset_probe.py)
  __shared__ unsigned long long red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];      // integer sums: exact in any order
}

__global__ __launch_bounds__(256) void k_sum_u64(const unsigned long long* __restrict__ in, const unsigned n,
                                                unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  unsigned long long acc = 0;
  for (unsigned i = threadIdx.x; i < n; i += 256) acc += in[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

// ---- SSIM (single:44-57) ---------------------------------------------------------------
// mean of  (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2))  with the 11 x 11, sigma 1.5 Gaussian moments.
// s1 and s2 only enter as their SUM, so FOUR blurred fields suffice instead of the reference's five: x, y, x^2 + y^2, xy.
//
// Form (72 -> 37 us per 4K plane over round 3): one wave per strip of 64 columns, one image COLUMN per
// lane, the wave walks down SS_R output rows.  Per input row: the row's (x, y, x^2 + y^2, xy) of the wave's 74 columns go to
// LDS as one float4 per pixel (two rows double-buffered; a single wave needs no barrier), every lane reads its 11
// neighbours (ds_read_b128, conflict free) and forms the horizontal sums of the two field PAIRS - one v_pk_fma_f32 per pair
// and tap, written as explicit 2-vectors - into a ring of the last 11 rows held in REGISTERS (the row loop is unrolled 11
// times, so the ring indices are static: no moves); the vertical sums of the row that just became complete are 22 more
// packed FMAs (skipped behind a wave-uniform branch while the ring fills), then the SSIM quotient.  Rows are addressed
// through one buffer resource per image (lane column = VGPR offset, row = SGPR offset).  Nothing is computed twice except
// the horizontal sums of the 10 halo rows (SS_R = 34: 1.29 x on half of the arithmetic) and nothing but the input row
// passes through LDS.  What binds it (in-kernel stamps, occupancy sweep, ablations): VALU issue - 77 instructions per row of
// which 48 are the packed FMAs - at about 80 % of the SIMD's cycles; a lone wave needs 725 cycles per row (a dependent chain).
#ifndef WM_SSIM_ROWS
#define WM_SSIM_ROWS 34
#endif
constexpr int SH = 5, SS_R = WM_SSIM_ROWS, SS_W = 64, SS_IW = SS_W + 2 * SH;   // SS_R + 10 input rows = 4 turns of the 11-row ring

__device__ __forceinline__ int reflect101(int i, const int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i %= period; if (i < 0) i += period;
  return i < n ? i : period - i;
}

struct GaussTaps { float w[11]; };

#ifndef WM_SSIM_WAVES
#define WM_SSIM_WAVES 4
#endif
#ifndef WM_SSIM_WPB
#define WM_SSIM_WPB 1
#endif
constexpr int SS_WPB = WM_SSIM_WPB;       // independent waves per workgroup (each its own strip and LDS rows; no barrier)
template <typename TA, typename TB, bool TINY>
__global__ __launch_bounds__(64 * SS_WPB) __attribute__((amdgpu_waves_per_eu(WM_SSIM_WAVES, WM_SSIM_WAVES))) void k_ssim(const TA* __restrict__ img1, const size_t s1,
                                            const TB* __restrict__ img2, const size_t s2, const int H,
                                            const int W, const GaussTaps taps, double* __restrict__ out) {
  __shared__ float4 rowbuf_all[SS_WPB][2][2 * SS_W];     // [0, 74): the row; lanes 10..63 park their (unused) halo value behind it: no branch
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float4 (*rowbuf)[2 * SS_W] = rowbuf_all[wv];
  const int strip = blockIdx.x * SS_WPB + wv;
  const int x0 = strip * SS_W, y0 = blockIdx.y * SS_R;
  if (x0 >= W) {                                         // wave-uniform: a strip past the image contributes 0
    if (lane == 0) out[(size_t)blockIdx.y * (gridDim.x * SS_WPB) + strip] = 0.0;
    return;
  }
  // source columns of this lane: main (x0 - 5 + lane) and, for lanes 0..9, the right halo (x0 + 59 + lane).
  // Columns / rows past the image (tile overhang) are clamped into range first: their outputs are masked out anyway.
  // Borders: one fold suffices unless the image is smaller than the halo (TINY: the general formula, its own instantiation
  // so that the division it needs stays out of the row loop of every other image).
  auto refl = [&](int i, const int n) -> int {
    if (TINY) return reflect101(i, n);
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
  };
  // per-lane column offsets (32 bit) next to a wave-uniform row base: the loads need no vector address arithmetic
  const unsigned gxm = (unsigned)refl(min(x0 - SH + lane, W + SH - 1), W);
  const unsigned gxh = (unsigned)refl(min(x0 - SH + SS_W + min(lane, 2 * SH - 1), W + SH - 1), W);   // lanes >= 10 repeat lane 9's column
  constexpr int n_in = SS_R + 2 * SH;                    // input rows a wave walks through: a multiple of 11, so that the
  static_assert(n_in % 11 == 0, "ring turns");            // unrolled body needs no exit (rows past the image are clamped, their outputs masked)
  auto row_of = [&](const int ir) { return refl(min(y0 - SH + ir, H + SH - 1), H); };
  // PF rows in flight ahead of the one being worked on (register ring with static indices, like the sums)
  float am[11], bm[11], ah[11], bh[11];
  // buffer loads: one resource per image, the lane's column as the VGPR offset, the row as the SGPR offset - no 64-bit
  // address arithmetic per row (34 -> 12 SALU and 4 fewer VALU instructions per row); planes are < 2 GiB (checked on the host)
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)img1, 0, (int)(((size_t)(H - 1) * s1 + W) * sizeof(TA)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)img2, 0, (int)(((size_t)(H - 1) * s2 + W) * sizeof(TB)), 0x00020000);
  const unsigned row1 = (unsigned)(s1 * sizeof(TA)), row2 = (unsigned)(s2 * sizeof(TB));
  auto ld1 = [&](const unsigned col, const unsigned roff) -> float {
    if constexpr (sizeof(TA) == 1) return (float)__builtin_amdgcn_raw_buffer_load_b8(rs1, col, roff, 0);
    else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, col * 4u, roff, 0));
  };
  auto ld2 = [&](const unsigned col, const unsigned roff) -> float {
    if constexpr (sizeof(TB) == 1) return (float)__builtin_amdgcn_raw_buffer_load_b8(rs2, col, roff, 0);
    else return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, col * 4u, roff, 0));
  };
#define SSIM_FETCH(slot, ir)                                                               \
  do {                                                                                     \
    const unsigned gy_ = (unsigned)row_of(ir);                                             \
    const unsigned o1_ = gy_ * row1, o2_ = gy_ * row2;                                     \
    am[slot] = ld1(gxm, o1_); bm[slot] = ld2(gxm, o2_);                                    \
    ah[slot] = ld1(gxh, o1_); bh[slot] = ld2(gxh, o2_);                                    \
  } while (0)
#ifndef WM_SSIM_PREFETCH
#define WM_SSIM_PREFETCH 3
#endif
  constexpr int PF = WM_SSIM_PREFETCH;                   // input rows in flight ahead of the one being worked on.  What matters is
#pragma unroll                                           // that NOTHING in the row loop branches per lane: with the halo loads and
  for (int r = 0; r < PF; ++r) SSIM_FETCH(r, r);         // stores under `if (lane < 10)` hipcc closed every block with vmcnt(0) and
                                                         // the prefetch was void (68 us per 4K plane, as slow as the tiled form)
  // the two field pairs (x, y) and (x^2 + y^2, xy) as explicit 2-vectors: every tap is one v_pk_fma_f32 per pair.  Left to
  // the SLP vectoriser, 24 of a row's 94 FMAs stayed scalar `v_fmac_f32 v, s, v` - which costs 1.9 ns per wave-instruction
  // with its SGPR tap, as much as a packed one that does two (tools/ubench_ssim_fma.hip, profiles/r03y_ubench_ssim_fma.log)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 ringA[11], ringB[11];
  float acc = 0.0f;
#ifdef WM_SSIM_STAMPS
  unsigned long long st_e = 0, st_d = 0, t_mid = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
#ifndef WM_SSIM_ROTPRIO
#define WM_SSIM_ROTPRIO 0                                // rows per priority step; 0 (default) = leave the priority alone
#endif
#if WM_SSIM_ROTPRIO
  const int prio_rank = (int)(((unsigned)blockIdx.y * gridDim.x * SS_WPB + blockIdx.x * SS_WPB + wv) >> 10);   // dispatch order / 1 024 SIMDs
#endif
  const float C1 = (0.01f * 255) * (0.01f * 255), C2 = (0.03f * 255) * (0.03f * 255);
  const bool col_ok = x0 + lane < W;
#pragma unroll 1
  for (int base = 0; base < n_in; base += 11) {
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      const int ir = base + s;
#ifdef WM_SSIM_STAMPS                                     // diagnostic build (tools/ssim_stamps_probe.py): s_memtime at the row top ...
      const unsigned long long t_top = __builtin_amdgcn_s_memtime();
      if (ir > 0) st_d += t_top - t_mid;
#endif
#if WM_SSIM_ROTPRIO
      // Diagnostic option (-DWM_SSIM_ROTPRIO=n).  The SIMD's arbiter prefers its OLDEST wave: the four waves of a
      // SIMD finish one after the other (40 / 50 / 67 / 90 thousand cycles, in-kernel stamps, profiles/r03y_ssim_stamps.log).
      // A priority that rotates over the dispatch-order ranks every n rows makes them finish together (55-78 thousand) - and
      // changes nothing in the kernel's time (five boxes, interleaved A/B): the SIMD is VALU-issue-bound either way.
#ifdef WM_SSIM_STATICPRIO
      switch (prio_rank & 3) {
#else
      switch ((ir / WM_SSIM_ROTPRIO + prio_rank) & 3) {
#endif
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
#endif
      SSIM_FETCH((s + PF) % 11, ir + PF);                  // rows past n_in: clamped, never used
      float4* rb = rowbuf[ir & 1];                         // 11 is odd: the parity of s alone flips with base
      {
        const float a = am[s], b = bm[s], c = ah[s], d = bh[s];
        rb[lane] = make_float4(a, b, fmaf(a, a, b * b), a * b);
        rb[SS_W + lane] = make_float4(c, d, fmaf(c, c, d * d), c * d);
      }
#if defined(WM_SSIM_SYNC)
      __syncthreads();
#else
      // one wave per workgroup: the LDS executes a wave's instructions in order, so this row's writes (all lanes) are
      // done before its reads issue; only the COMPILER must be kept from hoisting a read of rb[lane + k] above the write
      // of rb[lane] (different addresses per lane).  A scheduling barrier costs nothing; s_waitcnt + s_barrier did.
      __builtin_amdgcn_wave_barrier();
#endif
      f2 hA = {0.f, 0.f}, hB = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float4 v = rb[lane + k];
        const f2 w2 = {taps.w[k], taps.w[k]};
        hA = __builtin_elementwise_fma(w2, (f2){v.x, v.y}, hA);
        hB = __builtin_elementwise_fma(w2, (f2){v.z, v.w}, hB);
      }
      ringA[s] = hA; ringB[s] = hB;
#ifdef WM_SSIM_STAMPS                                     // ... and once the row's horizontal sums are in the ring
      asm volatile("" :: "v"(hA), "v"(hB));
      t_mid = __builtin_amdgcn_s_memtime();
      st_e += t_mid - t_top;
#endif
      if (ir >= 2 * SH) {                                  // wave-uniform (a scalar branch): the first 10 rows only fill the ring.  Rows ir - 10 .. ir are in the ring: output row y0 + ir - 10
        f2 mA = {0.f, 0.f}, mB = {0.f, 0.f};              // (mu1, mu2), (E[x^2 + y^2], E[xy])
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const int r = (s + 1 + k) % 11;                  // oldest row first; the taps are symmetric
          const f2 w2 = {taps.w[k], taps.w[k]};
          mA = __builtin_elementwise_fma(w2, ringA[r], mA);
          mB = __builtin_elementwise_fma(w2, ringB[r], mB);
        }
        const float m1 = mA.x, m2 = mA.y, e = mB.x, q = mB.y;
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float num = (2.0f * m12 + C1) * (2.0f * (q - m12) + C2);
        const float den = (m11 + m22 + C1) * ((e - m11 - m22) + C2) + 1e-12f;
        const int oy = y0 + ir - 2 * SH;
        const float v = num * __builtin_amdgcn_rcpf(den);
        acc += (col_ok && oy < H) ? v : 0.0f;
      }
    }
  }
#undef SSIM_FETCH
#ifdef WM_SSIM_STAMPS
  if (lane == 0 && (blockIdx.x % 12 == 5) && (blockIdx.y % 9 == 4))      // a sample of workgroups across the dispatch order
    printf("blk %2d %2d rank %d dur %llu E %llu D %llu\n", blockIdx.x, blockIdx.y, (int)(((unsigned)blockIdx.y * gridDim.x + blockIdx.x) >> 10),
           (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin), st_e / n_in, st_d / n_in);
#endif
  double accd = (double)acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) accd += __shfl_down(accd, o, 64);
  if (lane == 0) out[(size_t)blockIdx.y * (gridDim.x * SS_WPB) + strip] = accd;
}

__global__ void k_sum_f64(const double* __restrict__ in, const size_t n, const double scale,
                          double* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) acc += in[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// ---- min-max normalise + clip + uint8 (single:221-222) ------------------------------
// f2ord / ord2f / block_minmax / NormQ: wm_internal.h (shared with the routed unscramble, wm_route.hip)

// min/max in two stages without atomics: each block leaves one {lo, hi} pair in
// mm[2*block ..]; the consumer (k_normalize_u8) folds the n_part pairs again per block.
constexpr unsigned MINMAX_BLOCKS = 512;

__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ x, const size_t n, unsigned* __restrict__ mm) {
  unsigned lo = 0xffffffffu, hi = 0u;
  const size_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);          // hipMalloc'd planes are 16-byte aligned (checked on the host)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    const unsigned a = f2ord(v.x), b = f2ord(v.y), c = f2ord(v.z), d = f2ord(v.w);
    lo = min(min(lo, a), min(b, min(c, d))); hi = max(max(hi, a), max(b, max(c, d)));
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned o = f2ord(x[i]);
    lo = min(lo, o); hi = max(hi, o);
  }
  block_minmax(lo, hi);
  if (threadIdx.x == 0) { mm[2 * blockIdx.x] = lo; mm[2 * blockIdx.x + 1] = hi; }
}

__global__ __launch_bounds__(256) void k_normalize_u8(const float* __restrict__ x, const size_t n,
                                                     const unsigned* __restrict__ mm, const unsigned n_part,
                                                     const int do_norm, uint8_t* __restrict__ out) {
  unsigned ulo = 0xffffffffu, uhi = 0u;
  if (do_norm) {
    for (unsigned i = threadIdx.x; i < n_part; i += blockDim.x) { ulo = min(ulo, mm[2 * i]); uhi = max(uhi, mm[2 * i + 1]); }
    block_minmax(ulo, uhi);
  }
  const NormQ q(ord2f(ulo), ord2f(uhi), do_norm);
  const size_t n4 = (((uintptr_t)out & 3u) == 0) ? n / 4 : 0;     // 4 pixels per thread: one 16-byte load, one 4-byte store
  const float4* x4 = reinterpret_cast<const float4*>(x);
  unsigned* out4 = reinterpret_cast<unsigned*>(out);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    out4[i] = q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (uint8_t)q(x[i]);
}

inline unsigned grid_for(size_t work_items, unsigned block = 256, unsigned cap = 256 * 8) {
  const size_t g = (work_items + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

int color_dispatch(wm_ctx* ctx, int op, const uint8_t* in3, const uint8_t* plane_in, uint8_t* out3,
                   uint8_t* plane_out, size_t n_px) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_px == 0) return WM_OK;
  if (!in3) return set_err(WM_ERR_BADARG, "input is NULL");
  if ((((uintptr_t)in3 | (uintptr_t)plane_in | (uintptr_t)out3 | (uintptr_t)plane_out) & 15u) != 0)
    return set_err(WM_ERR_BADARG, "pixel buffers must be 16-byte aligned");
  const dim3 grid(grid_for(n_px / 16 + 1)), block(256);
  switch (op) {
    case BGR_TO_YCC: hipLaunchKernelGGL((k_color<BGR_TO_YCC>), grid, block, 0, ctx->stream, in3, plane_in, out3, plane_out, n_px); break;
    case YCC_TO_BGR: hipLaunchKernelGGL((k_color<YCC_TO_BGR>), grid, block, 0, ctx->stream, in3, plane_in, out3, plane_out, n_px); break;
    case BGR_TO_GRAY: hipLaunchKernelGGL((k_color<BGR_TO_GRAY>), grid, block, 0, ctx->stream, in3, plane_in, out3, plane_out, n_px); break;
    case BGR_TO_Y: hipLaunchKernelGGL((k_color<BGR_TO_Y>), grid, block, 0, ctx->stream, in3, plane_in, out3, plane_out, n_px); break;
    default: hipLaunchKernelGGL((k_color<Y_INTO_YCC_TO_BGR>), grid, block, 0, ctx->stream, in3, plane_in, out3, plane_out, n_px); break;
  }
  WM_HIP(hipGetLastError());
  return WM_OK;
}

GaussTaps make_taps() {     // cv2.getGaussianKernel(11, 1.5)
  GaussTaps g;
  double s = 0, v[11];
  for (int i = 0; i < 11; ++i) { const double x = i - 5.0; v[i] = exp(-(x * x) / (2 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i < 11; ++i) g.w[i] = (float)(v[i] / s);
  return g;
}

}  // namespace

// ---- keyed scramble / unscramble of the watermark plane (single:66-80) ---------------------------
// The permutation itself stays NumPy's PCG64 shuffle on the host (bit-exact by construction); only the two
// index passes run here:  _permute: dst[i] = src[idx[i]]   _unpermute: dst[idx[i]] = src[i]
template <typename T>
__global__ __launch_bounds__(256) void k_permute(const T* __restrict__ src, const int* __restrict__ idx,
                                                float* __restrict__ dst, const size_t n, const size_t plane_stride_src) {
  src += (size_t)blockIdx.y * plane_stride_src;
  dst += (size_t)blockIdx.y * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (float)src[idx[i]];
}
__global__ __launch_bounds__(256) void k_unpermute(const float* __restrict__ src, const int* __restrict__ idx,
                                                  float* __restrict__ dst, const size_t n) {
  src += (size_t)blockIdx.y * n;
  dst += (size_t)blockIdx.y * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[idx[i]] = src[i];
}

extern "C" {

int wm_bgr_to_ycrcb_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* ycrcb, size_t n_px) {
  if (!ycrcb && n_px) return set_err(WM_ERR_BADARG, "output is NULL");
  return color_dispatch(ctx, BGR_TO_YCC, bgr, nullptr, ycrcb, nullptr, n_px);
}
int wm_ycrcb_to_bgr_u8_dev(wm_ctx* ctx, const uint8_t* ycrcb, uint8_t* bgr, size_t n_px) {
  if (!bgr && n_px) return set_err(WM_ERR_BADARG, "output is NULL");
  return color_dispatch(ctx, YCC_TO_BGR, ycrcb, nullptr, bgr, nullptr, n_px);
}
int wm_bgr_to_gray_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* gray, size_t n_px) {
  if (!gray && n_px) return set_err(WM_ERR_BADARG, "output is NULL");
  return color_dispatch(ctx, BGR_TO_GRAY, bgr, nullptr, nullptr, gray, n_px);
}
int wm_bgr_to_y_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* y, size_t n_px) {
  if (!y && n_px) return set_err(WM_ERR_BADARG, "output is NULL");
  return color_dispatch(ctx, BGR_TO_Y, bgr, nullptr, nullptr, y, n_px);
}
int wm_replace_y_u8_dev(wm_ctx* ctx, const uint8_t* bgr, const uint8_t* y_new, uint8_t* bgr_out, size_t n_px) {
  if ((!y_new || !bgr_out) && n_px) return set_err(WM_ERR_BADARG, "NULL argument");
  return color_dispatch(ctx, Y_INTO_YCC_TO_BGR, bgr, y_new, bgr_out, nullptr, n_px);
}

// sum of squared differences (device scalar, exact integer); psnr = 20 log10(255 / sqrt(ssd / n))
int wm_sqdiff_u8_dev(wm_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, unsigned long long* ssd_dev) {
  if (!ctx || !ssd_dev) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(wmi::use_ctx(ctx));
  if (n == 0) { WM_HIP(hipMemsetAsync(ssd_dev, 0, sizeof(unsigned long long), ctx->stream)); return WM_OK; }
  if (!a || !b || (((uintptr_t)a | (uintptr_t)b) & 15u)) return set_err(WM_ERR_BADARG, "buffers must be non-NULL and 16-byte aligned");
  const unsigned nblk = grid_for(n / 16 + 1);
  WM_TRY(grow(ctx, &ctx->partials, &ctx->partials_bytes, (size_t)nblk * sizeof(unsigned long long), "squared-difference partial sums"));
  unsigned long long* part = (unsigned long long*)ctx->partials;
  hipLaunchKernelGGL(k_sqdiff_u8, dim3(nblk), dim3(256), 0, ctx->stream, a, b, n, part);
  hipLaunchKernelGGL(k_sum_u64, dim3(1), dim3(256), 0, ctx->stream, part, nblk, ssd_dev);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// mean SSIM of two planes; kind bit0: img1 is float32 (else uint8), bit1: img2 is float32
int wm_ssim_dev(wm_ctx* ctx, const void* img1, size_t stride1, const void* img2, size_t stride2, int H, int W,
                int kind, double* ssim_dev) {
  if (!ctx || !img1 || !img2 || !ssim_dev) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(wmi::use_ctx(ctx));
  if (H <= 0 || W <= 0) return set_err(WM_ERR_BADARG, "H and W must be positive");
  if (stride1 < (size_t)W || stride2 < (size_t)W) return set_err(WM_ERR_BADARG, "row stride < W");
  {   // the kernel addresses each plane through a buffer resource with 32-bit offsets
    const size_t e1 = (kind & 1) ? 4 : 1, e2 = (kind & 2) ? 4 : 1;
    if (((size_t)(H - 1) * stride1 + W) * e1 >= ((size_t)1 << 31) || ((size_t)(H - 1) * stride2 + W) * e2 >= ((size_t)1 << 31))
      return set_err(WM_ERR_BADARG, "SSIM planes must be smaller than 2 GiB");
  }
  const dim3 grid(((W + SS_W - 1) / SS_W + SS_WPB - 1) / SS_WPB, (H + SS_R - 1) / SS_R), block(64 * SS_WPB);
  const size_t nblk = (size_t)grid.x * SS_WPB * grid.y;
  WM_TRY(grow(ctx, &ctx->partials, &ctx->partials_bytes, (nblk + 1) * sizeof(double), "ssim partial sums"));
  double* part = (double*)ctx->partials;
  const GaussTaps taps = make_taps();
  const bool tiny = (H < 2 * SH + 2) || (W < 2 * SH + 2);
#define WM_LAUNCH_SSIM(TA_, TB_)                                                                              \
  do {                                                                                                        \
    if (tiny) hipLaunchKernelGGL((k_ssim<TA_, TB_, true>), grid, block, 0, ctx->stream, (const TA_*)img1, stride1, (const TB_*)img2, stride2, H, W, taps, part); \
    else hipLaunchKernelGGL((k_ssim<TA_, TB_, false>), grid, block, 0, ctx->stream, (const TA_*)img1, stride1, (const TB_*)img2, stride2, H, W, taps, part); \
  } while (0)
  switch (kind & 3) {
    case 0: WM_LAUNCH_SSIM(uint8_t, uint8_t); break;
    case 1: WM_LAUNCH_SSIM(float, uint8_t); break;
    case 2: WM_LAUNCH_SSIM(uint8_t, float); break;
    default: WM_LAUNCH_SSIM(float, float); break;
  }
#undef WM_LAUNCH_SSIM
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, ctx->stream, part, nblk, 1.0 / ((double)H * (double)W), ssim_dev);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// out = uint8(clip(normalize_minmax(x), 0, 255))  (or just the clip when do_norm == 0)
int wm_normalize_u8_dev(wm_ctx* ctx, const float* x, size_t n, int do_norm, uint8_t* out) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n == 0) return WM_OK;
  if (!x || !out) return set_err(WM_ERR_BADARG, "NULL argument");
  if (((uintptr_t)x & 15u) != 0) return set_err(WM_ERR_BADARG, "float plane must be 16-byte aligned");
  WM_TRY(grow(ctx, &ctx->partials, &ctx->partials_bytes, MINMAX_BLOCKS * 2 * sizeof(unsigned), "minmax"));
  unsigned* mm = (unsigned*)ctx->partials;       // per-block {min, max} in order-preserving uint form
  const unsigned n_part = grid_for(n / 4 + 1, 256, MINMAX_BLOCKS);
  if (do_norm) hipLaunchKernelGGL(k_minmax, dim3(n_part), dim3(256), 0, ctx->stream, x, n, mm);
  hipLaunchKernelGGL(k_normalize_u8, dim3(grid_for(n / 4 + 1)), dim3(256), 0, ctx->stream, x, n, mm, n_part, do_norm, out);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

static int check_perm_args(wm_ctx* ctx, const void* src, const int* idx, const float* dst, size_t n, int n_planes) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || n_planes > 65535) return set_err(WM_ERR_BADARG, "n_planes must be in 0..65535");
  if (n > 0x7fffffffull) return set_err(WM_ERR_BADARG, "more than 2^31 - 1 elements per plane (the index is int32)");
  if (n && n_planes && (!src || !idx || !dst)) return set_err(WM_ERR_BADARG, "NULL argument");
  return WM_OK;
}

int wm_permute_u8_f32_dev(wm_ctx* ctx, const uint8_t* src, const int* idx, float* dst, size_t n, int n_planes) {
  WM_TRY(check_perm_args(ctx, src, idx, dst, n, n_planes));
  if (n == 0 || n_planes == 0) return WM_OK;
  hipLaunchKernelGGL((k_permute<uint8_t>), dim3(grid_for(n), n_planes), dim3(256), 0, ctx->stream, src, idx, dst, n, n);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

int wm_permute_f32_dev(wm_ctx* ctx, const float* src, const int* idx, float* dst, size_t n, int n_planes) {
  WM_TRY(check_perm_args(ctx, src, idx, dst, n, n_planes));
  if (n == 0 || n_planes == 0) return WM_OK;
  if (src == dst) return set_err(WM_ERR_BADARG, "permute cannot run in place");
  hipLaunchKernelGGL((k_permute<float>), dim3(grid_for(n), n_planes), dim3(256), 0, ctx->stream, src, idx, dst, n, n);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

int wm_unpermute_f32_dev(wm_ctx* ctx, const float* src, const int* idx, float* dst, size_t n, int n_planes) {
  WM_TRY(check_perm_args(ctx, src, idx, dst, n, n_planes));
  if (n == 0 || n_planes == 0) return WM_OK;
  if (src == dst) return set_err(WM_ERR_BADARG, "unpermute cannot run in place");
  hipLaunchKernelGGL(k_unpermute, dim3(grid_for(n), n_planes), dim3(256), 0, ctx->stream, src, idx, dst, n);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// ---- host-pointer conveniences ------------------------------------------------------
static int stage(wm_ctx* ctx, size_t bytes, char** base) {
  WM_TRY(grow(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes + 4096, "scratch"));
  *base = (char*)ctx->scratch;
  return WM_OK;
}
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

int wm_color_u8(wm_ctx* ctx, int op, const uint8_t* in3, const uint8_t* plane_in, uint8_t* out3, uint8_t* plane_out,
                size_t n_px) {
  WM_TRY(wmi::use_ctx(ctx));
  if (op < 0 || op > 4) return set_err(WM_ERR_BADARG, "unknown colour op");
  if (n_px == 0) return WM_OK;
  char* b;
  WM_TRY(stage(ctx, 2 * up256(n_px * 3) + 2 * up256(n_px), &b));
  uint8_t* d_in3 = (uint8_t*)b; uint8_t* d_out3 = d_in3 + up256(n_px * 3);
  uint8_t* d_pin = d_out3 + up256(n_px * 3); uint8_t* d_pout = d_pin + up256(n_px);
  if (!in3) return set_err(WM_ERR_BADARG, "input is NULL");
  WM_HIP(hipMemcpyAsync(d_in3, in3, n_px * 3, hipMemcpyHostToDevice, ctx->stream));
  if (op == Y_INTO_YCC_TO_BGR) {
    if (!plane_in) return set_err(WM_ERR_BADARG, "plane_in is NULL");
    WM_HIP(hipMemcpyAsync(d_pin, plane_in, n_px, hipMemcpyHostToDevice, ctx->stream));
  }
  WM_TRY(color_dispatch(ctx, op, d_in3, d_pin, d_out3, d_pout, n_px));
  if (op == BGR_TO_GRAY || op == BGR_TO_Y) {
    if (!plane_out) return set_err(WM_ERR_BADARG, "plane_out is NULL");
    WM_HIP(hipMemcpyAsync(plane_out, d_pout, n_px, hipMemcpyDeviceToHost, ctx->stream));
  } else {
    if (!out3) return set_err(WM_ERR_BADARG, "out3 is NULL");
    WM_HIP(hipMemcpyAsync(out3, d_out3, n_px * 3, hipMemcpyDeviceToHost, ctx->stream));
  }
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

int wm_psnr_u8(wm_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, double* psnr_out) {
  if (!ctx || !psnr_out) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(wmi::use_ctx(ctx));
  if (n == 0) { *psnr_out = 99.0; return WM_OK; }
  if (!a || !b) return set_err(WM_ERR_BADARG, "NULL argument");
  char* base;
  WM_TRY(stage(ctx, 2 * up256(n) + 256, &base));
  uint8_t* da = (uint8_t*)base; uint8_t* db = da + up256(n);
  unsigned long long* d_ssd = (unsigned long long*)(db + up256(n));
  WM_HIP(hipMemcpyAsync(da, a, n, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipMemcpyAsync(db, b, n, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_sqdiff_u8_dev(ctx, da, db, n, d_ssd));
  unsigned long long ssd = 0;
  WM_HIP(hipMemcpyAsync(&ssd, d_ssd, sizeof(ssd), hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  const double mse = (double)ssd / (double)n;
  // single:38-42 computes in float32: mse as float, 99.0 below 1e-12
  *psnr_out = (mse <= 1e-12) ? 99.0 : 20.0 * log10(255.0 / fmax(sqrt(mse), 1e-12));
  return WM_OK;
}

int wm_ssim(wm_ctx* ctx, const void* img1, const void* img2, int H, int W, int kind, double* ssim_out) {
  if (!ctx || !img1 || !img2 || !ssim_out) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(wmi::use_ctx(ctx));
  if (H <= 0 || W <= 0) return set_err(WM_ERR_BADARG, "H and W must be positive");
  const size_t n = (size_t)H * W;
  const size_t b1 = n * ((kind & 1) ? 4 : 1), b2 = n * ((kind & 2) ? 4 : 1);
  char* base;
  WM_TRY(stage(ctx, up256(b1) + up256(b2) + 256, &base));
  char* d1 = base; char* d2 = d1 + up256(b1); double* d_s = (double*)(d2 + up256(b2));
  WM_HIP(hipMemcpyAsync(d1, img1, b1, hipMemcpyHostToDevice, ctx->stream));
  WM_HIP(hipMemcpyAsync(d2, img2, b2, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_ssim_dev(ctx, d1, (size_t)W, d2, (size_t)W, H, W, kind, d_s));
  double s = 0;
  WM_HIP(hipMemcpyAsync(&s, d_s, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  *ssim_out = s;
  return WM_OK;
}

int wm_normalize_u8(wm_ctx* ctx, const float* x, size_t n, int do_norm, uint8_t* out) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n == 0) return WM_OK;
  if (!x || !out) return set_err(WM_ERR_BADARG, "NULL argument");
  char* base;
  WM_TRY(stage(ctx, up256(n * 4) + up256(n), &base));
  float* dx = (float*)base; uint8_t* dout = (uint8_t*)(base + up256(n * 4));
  WM_HIP(hipMemcpyAsync(dx, x, n * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_TRY(wm_normalize_u8_dev(ctx, dx, n, do_norm, dout));
  WM_HIP(hipMemcpyAsync(out, dout, n, hipMemcpyDeviceToHost, ctx->stream));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  return WM_OK;
}

}  // extern "C"
