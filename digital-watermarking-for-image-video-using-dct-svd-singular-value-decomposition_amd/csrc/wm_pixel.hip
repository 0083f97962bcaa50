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

// generic over pixel groups: in3 = interleaved 3-channel input (16 px = 48 B), plane = single-channel
// input/output (16 B per group), out3 = interleaved 3-channel output
template <int OP>
__global__ __launch_bounds__(256) void k_color(const uint8_t* __restrict__ in3, const uint8_t* __restrict__ plane_in,
                                              uint8_t* __restrict__ out3, uint8_t* __restrict__ plane_out,
                                              const size_t n_px) {
  const size_t n_groups = n_px / 16;
  for (size_t gidx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < n_groups;
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
  unsigned long long acc = 0;
  const size_t n16 = n / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = reinterpret_cast<const uint4*>(a)[i], y = reinterpret_cast<const uint4*>(b)[i];
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = (int)((xs[w] >> (8 * k)) & 0xffu) - (int)((ys[w] >> (8 * k)) & 0xffu);
        s += (uint32_t)(d * d);
      }
    acc += s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = n16 * 16; i < n; ++i) { const int d = (int)a[i] - (int)b[i]; acc += (unsigned)(d * d); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);     // integer sum: order-independent, exact
}

// ---- SSIM (single:44-57): 5 Gaussian-blurred moments per pixel ----------------------
// 32x32 output tile per 256-thread workgroup, 42x42 halo tile in LDS, separable 11 taps.
// tile of 64 x 32 outputs per workgroup, 5-pixel halo: 74 x 42 inputs (1.52x read amplification)
constexpr int SH = 5, TX = 64, TY = 32, SWX = TX + 2 * SH, SWY = TY + 2 * SH;
constexpr int SPX = SWX + 3;                      // LDS pitch of the input tiles: 77 = 13 mod 32, so the 4 rows x 16 stride-4 strips a wave reads land on distinct banks
constexpr size_t SSIM_LDS_BYTES = (size_t)(2 * SWY * SPX + 5 * SWY * (TX + 1)) * sizeof(float);   // 80,136 B

__device__ __forceinline__ int reflect101(int i, const int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i %= period; if (i < 0) i += period;
  return i < n ? i : period - i;
}

struct GaussTaps { float w[11]; };

// Mean SSIM partial sums: the five 11-tap separable blurs (x, y, xx, yy, xy) fused in LDS.
//   load 74 x 42 inputs -> horizontal pass (42 rows x 16 strips of 4 outputs) -> vertical pass
//   (64 columns x 4 strips of 8 rows: every thread busy) + SSIM map -> one double per workgroup.
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void k_ssim(const TA* __restrict__ img1, const size_t s1,
                                             const TB* __restrict__ img2, const size_t s2, const int H,
                                             const int W, const GaussTaps taps, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float (*A)[SPX] = reinterpret_cast<float (*)[SPX]>(lds);
  float (*B)[SPX] = reinterpret_cast<float (*)[SPX]>(lds + SWY * SPX);
  float (*Hm)[SWY][TX + 1] = reinterpret_cast<float (*)[SWY][TX + 1]>(lds + 2 * SWY * SPX);   // [5][SWY][TX+1]
  __shared__ double red[4];
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  {
    // one input column per lane, rows ly = (t >> 7) + 2 i.  All 21 + 21 loads are issued before the
    // first LDS store (a load-store loop serialises on memory latency: 100 -> see DESIGN 7.1).
    // Borders: one reflection suffices unless the image is smaller than the halo (general formula).
    const int lx = t & 127;
    if (lx < SWX) {
      const bool tiny = (H < 2 * SH + 2) || (W < 2 * SH + 2);
      auto refl = [&](int i, const int n) -> int {
        if (tiny) return reflect101(i, n);
        i = i < 0 ? -i : i;
        return i >= n ? 2 * n - 2 - i : i;
      };
      // columns / rows past the image (tile overhang beyond W or H) may need a second fold; clamp the
      // source index into range first: those outputs are masked out of the sum anyway
      const int gx = refl(min(x0 + lx - SH, W + SH - 1), W);
      const TA* p1 = img1 + gx;
      const TB* p2 = img2 + gx;
      float va[SWY / 2], vb[SWY / 2];
#pragma unroll
      for (int i = 0; i < SWY / 2; ++i) {
        const int ly = (t >> 7) + 2 * i;
        const int gy = refl(min(y0 + ly - SH, H + SH - 1), H);
        va[i] = (float)p1[(size_t)gy * s1];
        vb[i] = (float)p2[(size_t)gy * s2];
      }
#pragma unroll
      for (int i = 0; i < SWY / 2; ++i) {
        const int ly = (t >> 7) + 2 * i;
        A[ly][lx] = va[i];
        B[ly][lx] = vb[i];
      }
    }
  }
  __syncthreads();
  // horizontal pass: SWY rows x 16 strips of 4 outputs, 14 loads per operand and work item.
  // The five running sums are kept as two packed pairs (x, y), (xx, yy) and one scalar (xy):
  // 3 instead of 5 FMAs per tap (the pass is VALU-issue-bound: 113 flop per pixel against 2 bytes).
  typedef float v2f __attribute__((ext_vector_type(2)));
  for (int e = t; e < SWY * (TX / 4); e += 256) {
    const int ly = e >> 4, lx0 = (e & 15) * 4;
    v2f ab1[14], ab2[14];
    float ab[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const float a = A[ly][lx0 + k], b = B[ly][lx0 + k];
      ab1[k] = v2f{a, b};
      ab2[k] = ab1[k] * ab1[k];
      ab[k] = a * b;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
      float sxy = 0;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float w = taps.w[k];
        const v2f wv = {w, w};
        s1 = __builtin_elementwise_fma(wv, ab1[j + k], s1);
        s2 = __builtin_elementwise_fma(wv, ab2[j + k], s2);
        sxy = fmaf(w, ab[j + k], sxy);
      }
      Hm[0][ly][lx0 + j] = s1.x; Hm[1][ly][lx0 + j] = s1.y; Hm[2][ly][lx0 + j] = s2.x;
      Hm[3][ly][lx0 + j] = s2.y; Hm[4][ly][lx0 + j] = sxy;
    }
  }
  __syncthreads();
  // vertical pass + SSIM map: thread = (column, strip of 8 rows); planes again as (x, y), (xx, yy), xy
  double acc = 0.0;
  const float C1 = (0.01f * 255) * (0.01f * 255), C2 = (0.03f * 255) * (0.03f * 255);
  {
    const int lx = t & 63, ly0 = (t >> 6) * 8;
    v2f m1[8], m2[8];
    float m5[8];
    {
      v2f c1[18], c2[18];
      float c5[18];
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        c1[k] = v2f{Hm[0][ly0 + k][lx], Hm[1][ly0 + k][lx]};
        c2[k] = v2f{Hm[2][ly0 + k][lx], Hm[3][ly0 + k][lx]};
        c5[k] = Hm[4][ly0 + k][lx];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v2f a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
        float a5 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const float w = taps.w[k];
          const v2f wv = {w, w};
          a1 = __builtin_elementwise_fma(wv, c1[j + k], a1);
          a2 = __builtin_elementwise_fma(wv, c2[j + k], a2);
          a5 = fmaf(w, c5[j + k], a5);
        }
        m1[j] = a1; m2[j] = a2; m5[j] = a5;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (y0 + ly0 + j >= H || x0 + lx >= W) continue;
      const float mu1 = m1[j].x, mu2 = m1[j].y;
      const float s1q = m2[j].x - mu1 * mu1, s2q = m2[j].y - mu2 * mu2, s12 = m5[j] - mu1 * mu2;
      const float num = (2 * mu1 * mu2 + C1) * (2 * s12 + C2);
      const float den = (mu1 * mu1 + mu2 * mu2 + C1) * (s1q + s2q + C2) + 1e-12f;
      acc += (double)(num / den);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
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
  WM_HIP(hipMemsetAsync(ssd_dev, 0, sizeof(unsigned long long), ctx->stream));
  if (n == 0) return WM_OK;
  if (!a || !b || (((uintptr_t)a | (uintptr_t)b) & 15u)) return set_err(WM_ERR_BADARG, "buffers must be non-NULL and 16-byte aligned");
  hipLaunchKernelGGL(k_sqdiff_u8, dim3(grid_for(n / 16 + 1)), dim3(256), 0, ctx->stream, a, b, n, ssd_dev);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// mean SSIM of two planes; kind bit0: img1 is float32 (else uint8), bit1: img2 is float32
int wm_ssim_dev(wm_ctx* ctx, const void* img1, size_t stride1, const void* img2, size_t stride2, int H, int W,
                int kind, double* ssim_dev) {
  if (!ctx || !img1 || !img2 || !ssim_dev) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(wmi::use_ctx(ctx));
  if (H <= 0 || W <= 0) return set_err(WM_ERR_BADARG, "H and W must be positive");
  const dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY), block(256);
  const size_t nblk = (size_t)grid.x * grid.y;
  WM_TRY(grow(ctx, &ctx->partials, &ctx->partials_bytes, (nblk + 1) * sizeof(double), "ssim partial sums"));
  double* part = (double*)ctx->partials;
  const GaussTaps taps = make_taps();
#define WM_LAUNCH_SSIM(TA_, TB_)                                                                              \
  do { /* 80 KB of dynamic LDS is above the 64 KB default cap: raise it (per device, cheap) */                \
    WM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ssim<TA_, TB_>),                              \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SSIM_LDS_BYTES));             \
    hipLaunchKernelGGL((k_ssim<TA_, TB_>), grid, block, SSIM_LDS_BYTES, ctx->stream, (const TA_*)img1,        \
                       stride1, (const TB_*)img2, stride2, H, W, taps, part);                                 \
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
