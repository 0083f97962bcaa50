// Routed (two-pass, fully coalesced) form of the keyed unscramble + min-max normalise at the end of the
// reference's extract (app_dct_svd_single.py:74-80 `_unpermute`, :221-222 cv2.normalize / clip / uint8):
//
//     out[idx[i]] = uint8(clip((w[i] - min w) * 255 / (max w - min w), 0, 255))            per plane
//
// idx is a uniformly random permutation of a whole plane (NumPy's PCG64 shuffle of arange(H*W), made on the host),
// so the literal index pass touches one 128-byte line per 4-byte element: measured 154 us per 4K plane for the
// scatter `dst[idx[i]] = src[i]` (646 GB/s of algorithmic bytes, 8 % of the HBM roof; profiles/r03_pixel_*).
// The permutation is fixed per key, so it is factored ONCE per key into two block-local permutations around a
// block transpose (a `wm_route`), after which every global access of the per-frame work is coalesced:
//
//   blocks of S = 2^15 elements; a = source block of i, b = destination block of idx[i]
//   pass 1 (one workgroup per source block a): read w coalesced, quantise (min / max are permutation invariant, so
//           they are taken on the scrambled plane), sort the block's S bytes by destination block in LDS (local
//           position l1[i], precomputed), write the runs of cell (a, b) to tmp - bucket-major, bucket b = exactly
//           the S slots of destination block b (idx is a bijection), runs ordered by a - as contiguous bytes;
//   pass 2 (one workgroup per destination block b): read the S bytes of bucket b coalesced, place each at its
//           offset inside the block (l2[g], precomputed) in LDS, write the block out with 16-byte stores.
//
// Traffic per pixel: min-max 4 B, pass 1 4 + 2 (l1) + 2 (bucket of the sorted slot) + 1, pass 2 1 + 2 + 1 = 17 B,
// all streaming, against 4 + 4 + 4 (scatter) + 4 (min-max) + 4 + 1 = 21 B of which 4 are random.
#include <string.h>

#include <vector>

#include "wm_internal.h"

using namespace wmi;

struct wm_route {
  size_t n = 0;              // elements per plane
  int log_s = 15, S = 1 << 15, nb = 0;
  void* base = nullptr;      // one device allocation holding everything below
  uint16_t* l1 = nullptr;    // [n]        pass 1: element i -> position of its byte in the sorted source block
  uint16_t* bkt = nullptr;   // [nb * S]   pass 1: destination block of sorted position t of source block a (a * S + t)
  uint16_t* l2 = nullptr;    // [n]        pass 2: tmp position g -> offset inside its destination block
  uint32_t* lrs = nullptr;   // [nb][nb + 1]  run starts inside the sorted source block a (prefix over b of cnt[a][b])
  uint32_t* cstart = nullptr;  // [nb][nb]    tmp position of the run of cell (a, b)
  int device = 0;
};

namespace {

constexpr int ROUTE_LOG_S = 15;
constexpr int ROUTE_NT = 1024;
constexpr int ROUTE_MAX_NB = 2048;        // run tables of a block live in LDS: 2 * 4 * nb bytes
constexpr unsigned MM_BLOCKS = 256;       // min-max partial pairs per plane

// ---- once per key ---------------------------------------------------------------------------------
// cnt[a][b] = elements of source block a that go to destination block b; l1[i] = arrival order of i inside its
// cell (any order will do: the two local permutations are built from the same numbers)
__global__ __launch_bounds__(ROUTE_NT) void k_route_count(const int* __restrict__ idx, const size_t n, const int log_s,
                                                         const int nb, uint32_t* __restrict__ cnt,
                                                         uint16_t* __restrict__ l1, unsigned* __restrict__ seen,
                                                         int* __restrict__ bad) {
  extern __shared__ uint32_t hist[];
  const int a = blockIdx.x, S = 1 << log_s;
  for (int b = threadIdx.x; b < nb; b += ROUTE_NT) hist[b] = 0;
  __syncthreads();
  const size_t base = (size_t)a << log_s;
  const int cnt_a = (int)min((size_t)S, n - base);
  for (int k = threadIdx.x; k < cnt_a; k += ROUTE_NT) {
    const int d = idx[base + k];
    if ((unsigned)d >= n) { *bad = 1; continue; }
    if (atomicOr(&seen[d >> 5], 1u << (d & 31)) & (1u << (d & 31))) *bad = 2;      // a destination named twice: not a bijection
    l1[base + k] = (uint16_t)atomicAdd(&hist[d >> log_s], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += ROUTE_NT) cnt[(size_t)a * nb + b] = hist[b];
}

__global__ __launch_bounds__(ROUTE_NT) void k_route_fill(const int* __restrict__ idx, const size_t n, const int log_s,
                                                        const int nb, const uint32_t* __restrict__ lrs,
                                                        const uint32_t* __restrict__ cstart, uint16_t* __restrict__ l1,
                                                        uint16_t* __restrict__ bkt, uint16_t* __restrict__ l2) {
  extern __shared__ uint32_t tab[];          // lrs row [nb + 1] | cstart row [nb]
  uint32_t* lr = tab; uint32_t* cs = tab + nb + 1;
  const int a = blockIdx.x, S = 1 << log_s;
  for (int b = threadIdx.x; b <= nb; b += ROUTE_NT) lr[b] = lrs[(size_t)a * (nb + 1) + b];
  for (int b = threadIdx.x; b < nb; b += ROUTE_NT) cs[b] = cstart[(size_t)a * nb + b];
  __syncthreads();
  const size_t base = (size_t)a << log_s;
  const int cnt_a = (int)min((size_t)S, n - base);
  for (int k = threadIdx.x; k < cnt_a; k += ROUTE_NT) {
    const int d = idx[base + k];
    const int b = d >> log_s, slot = l1[base + k];
    const uint32_t t = lr[b] + slot;
    l1[base + k] = (uint16_t)t;
    bkt[base + t] = (uint16_t)b;
    l2[(size_t)cs[b] + slot] = (uint16_t)(d & (S - 1));
  }
}

// ---- per frame ------------------------------------------------------------------------------------
// per-plane min / max as MM_BLOCKS {lo, hi} pairs in order-preserving uint form (no atomics: the consumer folds)
__global__ __launch_bounds__(256) void k_minmax_planes(const float* __restrict__ x, const size_t n, unsigned* __restrict__ mm) {
  x += (size_t)blockIdx.y * n;
  mm += (size_t)blockIdx.y * 2 * gridDim.x;
  unsigned lo = 0xffffffffu, hi = 0u;
  const bool al = (((uintptr_t)x) & 15u) == 0;
  const size_t n4 = al ? n / 4 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
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

__global__ __launch_bounds__(ROUTE_NT) void k_route_p1(const float* __restrict__ src, const size_t n, const int log_s,
                                                      const int nb, const uint16_t* __restrict__ l1,
                                                      const uint16_t* __restrict__ bkt, const uint32_t* __restrict__ lrs,
                                                      const uint32_t* __restrict__ cstart, const unsigned* __restrict__ mm,
                                                      const unsigned n_part, const int do_norm, const int include_zero,
                                                      uint8_t* __restrict__ tmp) {
  extern __shared__ uint32_t smem[];         // lrs row [nb + 1] | cstart row [nb] | bytes [S]
  uint32_t* lr = smem; uint32_t* cs = smem + nb + 1;
  uint8_t* val = reinterpret_cast<uint8_t*>(smem + 2 * nb + 2);
  const int a = blockIdx.x, S = 1 << log_s;
  src += (size_t)blockIdx.y * n; tmp += (size_t)blockIdx.y * n; mm += (size_t)blockIdx.y * 2 * n_part;
  for (int b = threadIdx.x; b <= nb; b += ROUTE_NT) lr[b] = lrs[(size_t)a * (nb + 1) + b];
  for (int b = threadIdx.x; b < nb; b += ROUTE_NT) cs[b] = cstart[(size_t)a * nb + b];
  unsigned ulo = 0xffffffffu, uhi = 0u;
  if (do_norm) {
    for (unsigned i = threadIdx.x; i < n_part; i += ROUTE_NT) { ulo = min(ulo, mm[2 * i]); uhi = max(uhi, mm[2 * i + 1]); }
    if (include_zero) { ulo = min(ulo, f2ord(0.0f)); uhi = max(uhi, f2ord(0.0f)); }
    block_minmax(ulo, uhi);                  // ends in a barrier
  } else {
    __syncthreads();
  }
  const NormQ q(ord2f(ulo), ord2f(uhi), do_norm);       // the same arithmetic as k_normalize_u8: identical bytes
  const size_t base = (size_t)a << log_s;
  const int cnt_a = (int)min((size_t)S, n - base);
  for (int k = threadIdx.x; k < cnt_a; k += ROUTE_NT) val[l1[base + k]] = (uint8_t)q(src[base + k]);
  __syncthreads();
  for (int t = threadIdx.x; t < cnt_a; t += ROUTE_NT) {
    const int b = bkt[base + t];
    tmp[(size_t)cs[b] + (t - lr[b])] = val[t];
  }
}

__global__ __launch_bounds__(ROUTE_NT) void k_route_p2(const uint8_t* __restrict__ tmp, const size_t n, const int log_s,
                                                      const uint16_t* __restrict__ l2, uint8_t* __restrict__ dst) {
  extern __shared__ uint32_t smem[];
  uint8_t* out = reinterpret_cast<uint8_t*>(smem);
  const int b = blockIdx.x, S = 1 << log_s;
  tmp += (size_t)blockIdx.y * n; dst += (size_t)blockIdx.y * n;
  const size_t base = (size_t)b << log_s;
  const int cnt = (int)min((size_t)S, n - base);
  for (int g = threadIdx.x; g < cnt; g += ROUTE_NT) out[l2[base + g]] = tmp[base + g];
  __syncthreads();
  uint8_t* o = dst + base;
  if ((((uintptr_t)o) & 15u) == 0) {
    const int n16 = cnt / 16;
    const uint4* s16 = reinterpret_cast<const uint4*>(out);
    uint4* o16 = reinterpret_cast<uint4*>(o);
    for (int t = threadIdx.x; t < n16; t += ROUTE_NT) o16[t] = s16[t];
    for (int t = n16 * 16 + threadIdx.x; t < cnt; t += ROUTE_NT) o[t] = out[t];
  } else {
    for (int t = threadIdx.x; t < cnt; t += ROUTE_NT) o[t] = out[t];
  }
}


// ---- the scramble direction through the same tables (single:66-72 `_permute`: dst[i] = src[idx[i]]) -----------
// pass A (one workgroup per block b of the SOURCE plane = destination block of the route): the block's bytes into LDS,
//         tmp[g] = block[l2[g]] for the slots g of bucket b - coalesced writes;
// pass B (one workgroup per block a of the scrambled plane): the runs of cell (a, b) from tmp into LDS in sorted order,
//         dst[i] = (float) sorted[l1[i]] - coalesced float stores.
__global__ __launch_bounds__(ROUTE_NT) void k_route_ga(const uint8_t* __restrict__ src, const size_t n, const int log_s,
                                                      const uint16_t* __restrict__ l2, uint8_t* __restrict__ tmp) {
  extern __shared__ uint32_t smem[];
  uint8_t* blk = reinterpret_cast<uint8_t*>(smem);
  const int b = blockIdx.x, S = 1 << log_s;
  src += (size_t)blockIdx.y * n; tmp += (size_t)blockIdx.y * n;
  const size_t base = (size_t)b << log_s;
  const int cnt = (int)min((size_t)S, n - base);
  const uint8_t* s = src + base;
  if ((((uintptr_t)s) & 15u) == 0) {
    const int n16 = cnt / 16;
    const uint4* s16 = reinterpret_cast<const uint4*>(s);
    uint4* b16 = reinterpret_cast<uint4*>(blk);
    for (int t = threadIdx.x; t < n16; t += ROUTE_NT) b16[t] = s16[t];
    for (int t = n16 * 16 + threadIdx.x; t < cnt; t += ROUTE_NT) blk[t] = s[t];
  } else {
    for (int t = threadIdx.x; t < cnt; t += ROUTE_NT) blk[t] = s[t];
  }
  __syncthreads();
  for (int g = threadIdx.x; g < cnt; g += ROUTE_NT) tmp[base + g] = blk[l2[base + g]];
}

__global__ __launch_bounds__(ROUTE_NT) void k_route_gb(const uint8_t* __restrict__ tmp, const size_t n, const int log_s,
                                                      const int nb, const uint16_t* __restrict__ l1,
                                                      const uint16_t* __restrict__ bkt, const uint32_t* __restrict__ lrs,
                                                      const uint32_t* __restrict__ cstart, float* __restrict__ dst) {
  extern __shared__ uint32_t smem[];         // lrs row [nb + 1] | cstart row [nb] | bytes [S]
  uint32_t* lr = smem; uint32_t* cs = smem + nb + 1;
  uint8_t* val = reinterpret_cast<uint8_t*>(smem + 2 * nb + 2);
  const int a = blockIdx.x, S = 1 << log_s;
  tmp += (size_t)blockIdx.y * n; dst += (size_t)blockIdx.y * n;
  for (int b = threadIdx.x; b <= nb; b += ROUTE_NT) lr[b] = lrs[(size_t)a * (nb + 1) + b];
  for (int b = threadIdx.x; b < nb; b += ROUTE_NT) cs[b] = cstart[(size_t)a * nb + b];
  __syncthreads();
  const size_t base = (size_t)a << log_s;
  const int cnt_a = (int)min((size_t)S, n - base);
  for (int t = threadIdx.x; t < cnt_a; t += ROUTE_NT) {
    const int b = bkt[base + t];
    val[t] = tmp[(size_t)cs[b] + (t - lr[b])];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cnt_a; k += ROUTE_NT) dst[base + k] = (float)val[l1[base + k]];
}

// ---- copy kernel between device memory and MAPPED pinned host memory (either direction) ---------------------------
// A frame pipeline keeps both PCIe directions busy with a copy per direction on its own stream.  On this stack
// hipMemcpyAsync chooses per call between an SDMA engine and a blit kernel (D2H: the blit kernel, nearly always), and
// in one of the resulting modes an H2D copy does not start before the previous batch's D2H has finished - the three
// stages serialise (1 715 frames/s against 2 455, tools/e2e_variants.py).  This kernel is that blit copy made explicit:
// a few workgroups (the link, not the CUs, is the limit), 16-byte accesses, four in flight per lane.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const v4u* __restrict__ src, v4u* __restrict__ dst, const size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const v4u a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
              c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ void k_copy_tail(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

inline size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

namespace wmi {
int route_unpermute_normalize(wm_ctx* ctx, const float* src, const wm_route* r, uint8_t* dst, size_t n, int n_planes, int do_norm,
                              const unsigned* mm_ext, unsigned n_part_ext, int include_zero) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || n_planes > 65535) return set_err(WM_ERR_BADARG, "n_planes must be in 0..65535");
  if (!r) return set_err(WM_ERR_BADARG, "route is NULL");
  if (r->n != n) return set_err(WM_ERR_BADARG, "the route was built for another plane size");
  if (r->device != ctx->device) return set_err(WM_ERR_BADARG, "the route lives on another device");
  if (n_planes == 0) return WM_OK;
  if (!src || !dst) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(grow(ctx, &ctx->route_tmp, &ctx->route_tmp_bytes, (size_t)n_planes * n + (size_t)n_planes * MM_BLOCKS * 8 + 256, "route staging"));
  uint8_t* tmp = (uint8_t*)ctx->route_tmp;
  const unsigned* mm = mm_ext;
  unsigned n_part = n_part_ext;
  if (do_norm && !mm_ext) {
    unsigned* mm_own = (unsigned*)((char*)ctx->route_tmp + a256((size_t)n_planes * n));
    n_part = (unsigned)std::min<size_t>(MM_BLOCKS, (n / 4 + 256) / 256);
    hipLaunchKernelGGL(k_minmax_planes, dim3(n_part, n_planes), dim3(256), 0, ctx->stream, src, n, mm_own);
    mm = mm_own;
  }
  const size_t lds1 = (size_t)(2 * r->nb + 2) * 4 + r->S, lds2 = r->S;
  hipLaunchKernelGGL(k_route_p1, dim3(r->nb, n_planes), dim3(ROUTE_NT), lds1, ctx->stream, src, n, r->log_s, r->nb, r->l1, r->bkt,
                     r->lrs, r->cstart, mm, n_part, do_norm, include_zero, tmp);
  hipLaunchKernelGGL(k_route_p2, dim3(r->nb, n_planes), dim3(ROUTE_NT), lds2, ctx->stream, tmp, n, r->log_s, r->l2, dst);
  WM_HIP(hipGetLastError());
  return WM_OK;
}
}  // namespace wmi

extern "C" {

int wm_route_create_dev(wm_ctx* ctx, const int* idx, size_t n, wm_route** route_out) {
  WM_TRY(wmi::use_ctx(ctx));
  if (!idx || !route_out) return set_err(WM_ERR_BADARG, "NULL argument");
  *route_out = nullptr;
  if (n == 0 || n > 0x7fffffffull) return set_err(WM_ERR_BADARG, "n must be in 1..2^31-1 (the index is int32)");
  const int S = 1 << ROUTE_LOG_S;
  const size_t nb = (n + S - 1) / S;
  if (nb > ROUTE_MAX_NB) return set_err(WM_ERR_BADARG, "plane too large for a route (more than 2048 blocks of 32768 elements)");
  wm_route* r = new (std::nothrow) wm_route;
  if (!r) return set_err(WM_ERR_NOMEM, "host allocation failed for %s", "route");
  r->n = n; r->log_s = ROUTE_LOG_S; r->S = S; r->nb = (int)nb; r->device = ctx->device;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += a256(bytes); return o; };
  const size_t o_l1 = take(n * 2), o_bkt = take(nb * S * 2), o_l2 = take(n * 2), o_lrs = take(nb * (nb + 1) * 4),
               o_cs = take(nb * nb * 4), o_cnt = take(nb * nb * 4), o_bad = take(256), o_seen = take((n + 31) / 32 * 4);
  if (hipMalloc(&r->base, off) != hipSuccess) {
    (void)hipGetLastError();
    delete r;
    return set_err(WM_ERR_NOMEM, "hipMalloc failed for %s", "route tables");
  }
  char* b = (char*)r->base;
  r->l1 = (uint16_t*)(b + o_l1); r->bkt = (uint16_t*)(b + o_bkt); r->l2 = (uint16_t*)(b + o_l2);
  r->lrs = (uint32_t*)(b + o_lrs); r->cstart = (uint32_t*)(b + o_cs);
  uint32_t* d_cnt = (uint32_t*)(b + o_cnt); int* d_bad = (int*)(b + o_bad); unsigned* d_seen = (unsigned*)(b + o_seen);
  auto fail = [&](int rc) { (void)hipFree(r->base); delete r; return rc; };
#define WM_RT(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(set_err(WM_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_))); } while (0)
  WM_RT(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  WM_RT(hipMemsetAsync(d_seen, 0, (n + 31) / 32 * 4, ctx->stream));
  hipLaunchKernelGGL(k_route_count, dim3((unsigned)nb), dim3(ROUTE_NT), nb * 4, ctx->stream, idx, n, ROUTE_LOG_S, (int)nb, d_cnt, r->l1, d_seen, d_bad);
  WM_RT(hipGetLastError());
  std::vector<uint32_t> cnt(nb * nb), lrs(nb * (nb + 1)), cs(nb * nb);
  int bad = 0;
  WM_RT(hipMemcpyAsync(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  WM_RT(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  WM_RT(hipStreamSynchronize(ctx->stream));
  if (bad == 1) return fail(set_err(WM_ERR_BADARG, "index entries must lie in [0, n)"));
  if (bad) return fail(set_err(WM_ERR_BADARG, "idx is not a permutation of 0..n-1"));
  // a bijection sends exactly |block b| elements to every destination block b
  for (size_t bb = 0; bb < nb; ++bb) {
    size_t col = 0;
    for (size_t a = 0; a < nb; ++a) {
      cs[a * nb + bb] = (uint32_t)(bb * S + col);          // bucket b of tmp = the slots of destination block b
      col += cnt[a * nb + bb];
    }
    const size_t want = std::min((size_t)S, n - bb * S);
    if (col != want) return fail(set_err(WM_ERR_BADARG, "idx is not a permutation of 0..n-1"));
  }
  for (size_t a = 0; a < nb; ++a) {
    uint32_t run = 0;
    for (size_t bb = 0; bb < nb; ++bb) { lrs[a * (nb + 1) + bb] = run; run += cnt[a * nb + bb]; }
    lrs[a * (nb + 1) + nb] = run;
  }
  WM_RT(hipMemcpyAsync(r->lrs, lrs.data(), lrs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  WM_RT(hipMemcpyAsync(r->cstart, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_route_fill, dim3((unsigned)nb), dim3(ROUTE_NT), (2 * nb + 1) * 4, ctx->stream, idx, n, ROUTE_LOG_S, (int)nb,
                     r->lrs, r->cstart, r->l1, r->bkt, r->l2);
  WM_RT(hipGetLastError());
  WM_RT(hipStreamSynchronize(ctx->stream));      // lrs / cs are locals
#undef WM_RT
  *route_out = r;
  return WM_OK;
}

int wm_route_destroy(wm_ctx* ctx, wm_route* r) {
  if (!r) return WM_OK;
  WM_TRY(wmi::use_ctx(ctx));
  WM_HIP(hipStreamSynchronize(ctx->stream));
  if (r->base) (void)hipFree(r->base);
  delete r;
  return WM_OK;
}

int wm_unpermute_normalize_u8_dev(wm_ctx* ctx, const float* src, const wm_route* r, uint8_t* dst, size_t n, int n_planes,
                                  int do_norm) {
  return wmi::route_unpermute_normalize(ctx, src, r, dst, n, n_planes, do_norm, nullptr, 0, 0);
}

int wm_permute_u8_f32_routed_dev(wm_ctx* ctx, const uint8_t* src, const wm_route* r, float* dst, size_t n, int n_planes) {
  WM_TRY(wmi::use_ctx(ctx));
  if (n_planes < 0 || n_planes > 65535) return set_err(WM_ERR_BADARG, "n_planes must be in 0..65535");
  if (!r) return set_err(WM_ERR_BADARG, "route is NULL");
  if (r->n != n) return set_err(WM_ERR_BADARG, "the route was built for another plane size");
  if (r->device != ctx->device) return set_err(WM_ERR_BADARG, "the route lives on another device");
  if (n_planes == 0) return WM_OK;
  if (!src || !dst) return set_err(WM_ERR_BADARG, "NULL argument");
  WM_TRY(grow(ctx, &ctx->route_tmp, &ctx->route_tmp_bytes, (size_t)n_planes * n + 256, "route staging"));
  uint8_t* tmp = (uint8_t*)ctx->route_tmp;
  hipLaunchKernelGGL(k_route_ga, dim3(r->nb, n_planes), dim3(ROUTE_NT), (size_t)r->S, ctx->stream, src, n, r->log_s, r->l2, tmp);
  hipLaunchKernelGGL(k_route_gb, dim3(r->nb, n_planes), dim3(ROUTE_NT), (size_t)(2 * r->nb + 2) * 4 + r->S, ctx->stream, tmp, n,
                     r->log_s, r->nb, r->l1, r->bkt, r->lrs, r->cstart, dst);
  WM_HIP(hipGetLastError());
  return WM_OK;
}

// dst / src: one of them device memory, the other pinned host memory that is mapped into the device's address space
// (hipHostMalloc / hipHostRegister - torch's pin_memory() is); enqueued on the context's stream, no synchronisation
int wm_copy_mapped_dev(wm_ctx* ctx, void* dst, const void* src, size_t bytes, int n_workgroups) {
  WM_TRY(wmi::use_ctx(ctx));
  if (bytes == 0) return WM_OK;
  if (!dst || !src) return set_err(WM_ERR_BADARG, "NULL argument");
  void* d_dst = dst; void* d_src = const_cast<void*>(src);
  // Only memory the kernel can dereference is accepted: device or managed allocations, and host allocations that are mapped
  // into the device's address space.  ROCm answers hipSuccess with hipMemoryTypeUnregistered for ordinary pageable host memory
  // (a kernel touching it faults), so every other type is refused; the range must also lie inside its allocation.
  hipPointerAttribute_t at;
  for (void** pp : {&d_dst, &d_src}) {
    if (hipPointerGetAttributes(&at, *pp) != hipSuccess) { (void)hipGetLastError(); return set_err(WM_ERR_BADARG, "pointer is neither device nor pinned host memory"); }
    if (at.type == hipMemoryTypeHost) {
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, *pp, 0) != hipSuccess || !dp) { (void)hipGetLastError(); return set_err(WM_ERR_BADARG, "host memory is not mapped for the device"); }
      *pp = dp;
    } else if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) {
      return set_err(WM_ERR_BADARG, "pointer is neither device nor pinned host memory");
    }
    void* base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)*pp) == hipSuccess && base && size) {
      if ((const char*)*pp + bytes > (const char*)base + size) return set_err(WM_ERR_BADARG, "the copy runs past the end of its allocation");
    } else {
      (void)hipGetLastError();             // no range known for this pointer (some host registrations): the type check above stands
    }
  }
  if (n_workgroups <= 0) n_workgroups = 64;
  if (n_workgroups > 4096) n_workgroups = 4096;
  size_t head = 0;
  if (((((uintptr_t)d_dst) | ((uintptr_t)d_src)) & 15u) == 0) {
    const size_t n16 = bytes / 16;
    if (n16) hipLaunchKernelGGL(k_copy16, dim3(n_workgroups), dim3(256), 0, ctx->stream, (const v4u*)d_src, (v4u*)d_dst, n16);
    head = n16 * 16;
  }
  if (head < bytes) {
    const size_t rest = bytes - head;
    hipLaunchKernelGGL(k_copy_tail, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)d_src + head, (uint8_t*)d_dst + head, rest);
  }
  WM_HIP(hipGetLastError());
  return WM_OK;
}

}  // extern "C"
