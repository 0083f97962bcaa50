// Internal declarations shared by the translation units of libwmhip.so
// (wmhip.hip: tile-mode kernels + context; wm_ref.hip: full-frame mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/wmhip.h"

namespace wmi {

constexpr int WAVE = 64;
constexpr int N_EVENTS = 64;

int set_err(int code, const char* fmt, const char* a = "", const char* b = "");

#define WM_HIP(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return wmi::set_err(WM_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define WM_TRY(call)              \
  do {                            \
    int rc_ = (call);             \
    if (rc_ != WM_OK) return rc_; \
  } while (0)

}  // namespace wmi

struct wm_ctx {
  int device = 0;
  int n_cu = 256;                 // multiProcessorCount of the device (persistent grids are sized from it)
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int* d_status = nullptr;        // [0] sticky kernel status, [1] embed literal-fallback count, [2] constant-tile count
  void* scratch = nullptr;        // grow-only device scratch (host-pointer wrappers)
  size_t scratch_bytes = 0;
  void* partials = nullptr;       // grow-only detect partial sums
  size_t partials_bytes = 0;
  void* fb_list = nullptr;        // grow-only list of tiles for the embed fallback
  size_t fb_bytes = 0;
  void* ref_ws = nullptr;         // grow-only workspace of the full-frame mode
  size_t ref_ws_bytes = 0;
  void* ref_ws2 = nullptr;        // second one: null-space completion of rank-deficient planes
  size_t ref_ws2_bytes = 0;
  void* route_tmp = nullptr;      // grow-only staging of the routed unscramble (wm_route.hip): bucket-major bytes + min-max pairs
  size_t route_tmp_bytes = 0;
  void* extract_f32 = nullptr;    // grow-only float estimate + min-max partials of wm_extract_unscrambled_u8_dev
  size_t extract_f32_bytes = 0;
  static constexpr int MAX_PAIR_TABS = 6;   // round-robin tournaments of the block Jacobi, by block count
  void* pair_tab[MAX_PAIR_TABS] = {};
  int pair_tab_nbk[MAX_PAIR_TABS] = {};
  int pair_tab_next = 0;
  void* hier_dev[MAX_PAIR_TABS] = {};       // two-level (super-block) tournament tables of the block Jacobi, by block count: device part
  void* hier_host[MAX_PAIR_TABS] = {};      // ... and the host part (malloc)
  int hier_key[MAX_PAIR_TABS] = {};         // block count * 16 + super-block size the table was built for
  int hier_next = 0;
  void* hier_ws = nullptr;        // grow-only workspace of the two-level scheme (tracked Gram matrices, rotations, partial sums)
  size_t hier_ws_bytes = 0;
  float* dct_mat[2] = {nullptr, nullptr};   // cached DCT-II basis matrices (device), by size
  int dct_n[2] = {0, 0};
  int ref_last_sweeps = 0;        // outer Jacobi sweeps of the last full-frame SVD (diagnostics)
  double ref_last_flops = 0.0;    // matrix-core flops its Gram / rotation products issued (nominal: skipped pairs counted)
  int ref_last_hier = 0;          // 1: it ran the two-level scheme
  float ref_skip_thr = 0.0f;      // residual cosine the last full-frame Jacobi may have left between two rows
  static constexpr int MAX_AUX = 7;   // extra queues of the batched full-frame Jacobi (created on first use)
  hipStream_t aux_stream[MAX_AUX] = {};
  hipEvent_t ev_fork[MAX_AUX] = {}, ev_join[MAX_AUX] = {};
  hipEvent_t ev[wmi::N_EVENTS] = {};
};

namespace wmi {
// first statement of every entry point that takes a context: NULL check, and make the
// context's device current (a process may hold contexts on several devices; hipMalloc and
// kernel launches follow the calling thread's current device)
inline int use_ctx(const wm_ctx* ctx) {
  if (!ctx) return set_err(WM_ERR_BADARG, "ctx is NULL");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return set_err(WM_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  return WM_OK;
}
// grow-only device buffer (synchronises the stream before freeing the old one)
int grow(wm_ctx* ctx, void** buf, size_t* have, size_t bytes, const char* what);
// wm_ref.hip: releases the host part of a two-level tournament table (wm_ctx::hier_host)
void hier_host_free(void* tab);
// wm_route.hip: routed unscramble + normalise; mm_ext != NULL: n_part_ext {min, max} pairs per plane already on the device
// (order-preserving uint form), include_zero: the value 0 also takes part in the min / max
int route_unpermute_normalize(wm_ctx* ctx, const float* src, const wm_route* r, uint8_t* dst, size_t n, int n_planes, int do_norm,
                              const unsigned* mm_ext, unsigned n_part_ext, int include_zero);

// ---- min-max normalise + clip + uint8 (single:221-222), shared by wm_pixel.hip and wm_route.hip ----
__device__ __forceinline__ unsigned f2ord(float f) {   // order-preserving float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
// block-wide min / max of order-preserving uints (any block size up to 1024); ends in a barrier, every thread
// returns with the block's values
__device__ __forceinline__ void block_minmax(unsigned& lo, unsigned& hi) {
  __shared__ unsigned s_lo[16], s_hi[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_down(lo, o, 64)); hi = max(hi, __shfl_down(hi, o, 64)); }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  lo = s_lo[0]; hi = s_hi[0];
  for (int w = 1; w < nw; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
}
// cv2.normalize(x, None, 0, 255, NORM_MINMAX) followed by clip + truncate; do_norm == 0: clip + truncate only
struct NormQ {
  float lo, scale; int do_norm;
  __device__ NormQ(float lo_, float hi_, int dn) : lo(lo_), do_norm(dn) {
    const double range = (double)hi_ - (double)lo_;
    scale = (dn && range > 2.220446049250313e-16) ? (float)(255.0 / range) : 0.0f;
  }
  __device__ __forceinline__ unsigned operator()(float v) const {
    if (do_norm) v = (v - lo) * scale;
    return (unsigned)fminf(fmaxf(v, 0.0f), 255.0f);
  }
};
}  // namespace wmi
