/* wmhip.h - C ABI of the MI355X (gfx950) DCT-SVD watermark hot path.
 *
 * This is the drop-in boundary.  The reference
 * (Thitrongdan202/Digital-Watermarking-...-DCT-SVD) has no FFI seam: its hot
 * path is ~12 inline NumPy/OpenCV statements per branch inside
 * embed/extract/detect of app_dct_svd_single.py.  Each entry point below names
 * the reference statements (file:line, "single" = app_dct_svd_single.py,
 * "core" = dct_svd_core_secure.py) it replaces when those are applied to every
 * 8x8 tile of a plane ("tile-mode").  The binding a
 * maintainer would add on the reference side is a ctypes stub - see
 * INTEGRATION.md.
 *
 * Conventions
 *  - plain C, caller-owned buffers, nothing retained between calls;
 *  - every function returns an int status (WM_OK == 0); wm_last_error() gives
 *    the message for the calling thread;
 *  - a wm_ctx binds one device + one HIP stream; two contexts are independently
 *    usable from two threads;
 *  - *_dev entry points take DEVICE pointers, enqueue on the context's stream
 *    and return without synchronising; the un-suffixed ones take HOST pointers
 *    and do H2D + kernels + D2H + sync themselves;
 *  - planes are row-major uint8 (or float32) with `row_stride` in ELEMENTS
 *    between rows and `plane_stride` in ELEMENTS between planes; tiles are
 *    numbered t = ty * (W/8) + tx; per-tile arrays are tile-major:
 *    sigma[plane][t][8], U[plane][t][8][8] (row, col), Vt[plane][t][8][8].
 *  - rows/columns beyond the last full tile (H % 8, W % 8) are passed through
 *    unchanged by embed and written as 0 by extract/reconstruct.
 */
#ifndef WMHIP_H
#define WMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM_OK 0
#define WM_ERR_BADARG 1   /* NULL pointer, non-positive size, K outside 0..8 ...      */
#define WM_ERR_HIP 2      /* a HIP runtime call failed; wm_last_error() has the text  */
#define WM_ERR_NOCONV 3   /* Jacobi hit its sweep bound (numpy: LinAlgError)          */
#define WM_ERR_NOMEM 4    /* device or host allocation failed                         */

#define WM_TILE 8
#define WM_ABI_VERSION 1

typedef struct wm_ctx wm_ctx;

/* ---- library / context ------------------------------------------------- */
int wm_abi_version(void);
const char* wm_last_error(void);
int wm_device_count(int* n_out);
/* stream == NULL: the context creates (and owns) a non-blocking stream.
 * Otherwise `stream` is a hipStream_t the caller owns (e.g. torch's current
 * stream handle) and all work is enqueued there. */
int wm_create(int device, void* stream, wm_ctx** ctx_out);
int wm_destroy(wm_ctx* ctx);
int wm_sync(wm_ctx* ctx);
/* Reads-and-clears the sticky kernel status word (synchronises the stream):
 * WM_OK or WM_ERR_NOCONV. */
int wm_check_status(wm_ctx* ctx);

/* device memory + copies for callers that do not bring their own allocator */
int wm_malloc(wm_ctx* ctx, size_t bytes, void** dptr_out);
int wm_free(wm_ctx* ctx, void* dptr);
int wm_memcpy_h2d(wm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int wm_memcpy_d2h(wm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int wm_memset(wm_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* Copy kernel between device memory and pinned host memory that is mapped for the device (hipHostMalloc /
 * hipHostRegister; either direction), enqueued on the context's stream without synchronising: the explicit form of the
 * blit copy hipMemcpyAsync falls back to, for frame pipelines that keep one stream per PCIe direction busy.
 * n_workgroups <= 0: 64. */
int wm_copy_mapped_dev(wm_ctx* ctx, void* dst, const void* src, size_t bytes, int n_workgroups);

/* HIP-event timing on the context's stream (slot 0..63) */
int wm_event_record(wm_ctx* ctx, int slot);
int wm_event_elapsed_ms(wm_ctx* ctx, int slot_start, int slot_stop, float* ms_out);

/* ---- K1: fused tile-mode embed ------------------------------------------
 * Replaces, per 8x8 tile:  .astype(float32) (single:24,122) -> dct2
 * (single:32-33,172) -> np.linalg.svd (single:172) -> S_[:K] = Sc[:K] +
 * alpha*Sw[:K] (single:174-175) -> Uc @ diag(S_) @ Vct (single:176) -> idct2
 * (single:177) -> np.clip(.,0,255).astype(uint8) (single:27,145-147).
 *   host     [n_planes] uint8 planes (in)
 *   sigma_w  watermark singular values [.. n_tiles][8]; plane p reads
 *            sigma_w + p * sigma_w_plane_stride (0 => one set shared by all
 *            planes: the "watermark S computed once per video" shape)
 *   stego    [n_planes] uint8 planes (out; may alias host)
 *   sigma_c  [n_planes][n_tiles][8] host singular values "Sc" (out, for meta)
 *   yw       optional [n_planes][H][W] float32 unclipped stego (what gray-mode
 *            SSIM consumes, single:190); NULL to skip
 *   K        number of leading singular values perturbed, 0..8
 *            (reference: K = max(8, int(kfrac*8)) = 8). */
int wm_embed_tiles_u8_dev(wm_ctx* ctx, const uint8_t* host, const float* sigma_w,
                          uint8_t* stego, float* sigma_c, float* yw,
                          int n_planes, int H, int W, int row_stride, size_t plane_stride,
                          size_t sigma_w_plane_stride, float alpha, int K);
int wm_embed_tiles_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w,
                      uint8_t* stego, float* sigma_c, float* yw,
                      int n_planes, int H, int W, int row_stride, size_t plane_stride,
                      size_t sigma_w_plane_stride, float alpha, int K);

/* ---- K2: singular values of every tile of a uint8 plane ------------------
 * Replaces  Cw = dct2(Y); _, S_cw, _ = np.linalg.svd(Cw)  (single:205,
 * 234-236, 297, 305-307) per tile.  sigma: [n_planes][n_tiles][8]. */
int wm_sigma_tiles_u8_dev(wm_ctx* ctx, const uint8_t* planes, float* sigma,
                          int n_planes, int H, int W, int row_stride, size_t plane_stride);
int wm_sigma_tiles_u8(wm_ctx* ctx, const uint8_t* planes, float* sigma,
                      int n_planes, int H, int W, int row_stride, size_t plane_stride);

/* ---- K3: full SVD of every tile of a float32 plane (watermark side) ------
 * Replaces  Wm = dct2(wy_s); Uw, Sw, Vwt = np.linalg.svd(Wm)  (single:173,
 * 131-134) per tile.  U, Vt: [n_planes][n_tiles][8][8]; S: [..][8]. */
int wm_svd_tiles_f32_dev(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt,
                         int n_planes, int H, int W, int row_stride, size_t plane_stride);
int wm_svd_tiles_f32(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt,
                     int n_planes, int H, int W, int row_stride, size_t plane_stride);

/* ---- K2+K4: fused tile-mode extract --------------------------------------
 * Replaces, per tile:  dct2 + svd of the stego (single:205) ->
 * Sw_hat = (S_cw - Sc) / max(alpha,1e-8); Sw_hat[K:] = 0 (single:212-213) ->
 * Uw @ diag(Sw_hat) @ Vwt (single:214) -> idct2 (single:218).
 *   sigma_c  [n_planes][n_tiles][8]
 *   Uw, Vwt  [..][n_tiles][8][8]; plane p reads tile index p * uv_plane_stride + t,
 *            uv_plane_stride in TILES: n_tiles (per plane) or 0 (shared by all planes)
 *   out      [n_planes][H][W] float32 scrambled-watermark estimate wy_s
 *            (dense, row stride W) - input of _unpermute (single:220). */
int wm_extract_tiles_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c,
                            const float* Uw, const float* Vwt, float* out,
                            int n_planes, int H, int W, int row_stride, size_t plane_stride,
                            size_t uv_plane_stride, float alpha, int K);

/* Per-watermark preparation for wm_extract_tiles_px_u8_dev: the orthonormal DCT moves into
 * the factors, idct2(Uw diag(s) Vwt) = (D^T Uw) diag(s) (Vwt D) per 8x8 tile:
 *   Ux = D^T Uw,  Vxt = Vwt D   (n_tiles matrices of 8x8 each; in place allowed).
 * No reference counterpart: single:214-218 recomputes the product and the IDCT per call. */
int wm_tile_factors_to_pixel_dev(wm_ctx* ctx, const float* Uw, const float* Vwt, float* Ux, float* Vxt,
                                 size_t n_tiles);

/* wm_extract_tiles_u8_dev with the factors already in the pixel domain (above): same
 * result up to float32 rounding, without the per-tile IDCT (frames of a clip share one
 * watermark, so the preparation is paid once). */
int wm_extract_tiles_px_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Ux,
                               const float* Vxt, float* out, int n_planes, int H, int W, int row_stride,
                               size_t plane_stride, size_t uv_plane_stride, float alpha, int K);
int wm_extract_tiles_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c,
                        const float* Uw, const float* Vwt, float* out,
                        int n_planes, int H, int W, int row_stride, size_t plane_stride,
                        size_t uv_plane_stride, float alpha, int K);

/* The same, returning the SUM over the n_planes estimates ([H][W] float32, planes added in
 * ascending order on the device): the frames of a clip carry one watermark and are averaged by
 * the video extract, so one plane instead of n_planes crosses PCIe. */
int wm_extract_tiles_sum_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                            const float* Vwt, float* out_sum, int n_planes, int H, int W, int row_stride,
                            size_t plane_stride, size_t uv_plane_stride, float alpha, int K);

/* ---- K4: Uw diag(sw_hat) Vwt + idct2 per tile (single:214-218) ----------- */
int wm_reconstruct_tiles_dev(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt,
                             float* out, int n_planes, int H, int W);
int wm_reconstruct_tiles(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt,
                         float* out, int n_planes, int H, int W);

/* ---- K2+K5: fused tile-mode detect ---------------------------------------
 * Replaces  svd of the stego (single:297) -> Sw_hat = (S_cw - Sc)/max(alpha,
 * 1e-8) over ALL singular values (single:300) -> _nc(Sw, Sw_hat)
 * (single:284-289, 301) with the vectors flattened over all tiles.
 *   scores   [n_planes] float64 normalised-correlation score per plane
 *            (device pointer for _dev).  The caller applies  score >= thresh
 *            and the 3-plane mean of colour mode (single:302, 317-318). */
int wm_detect_tiles_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c,
                           const float* sigma_w, double* scores,
                           int n_planes, int H, int W, int row_stride, size_t plane_stride,
                           size_t sigma_w_plane_stride, float alpha);
int wm_detect_tiles_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c,
                       const float* sigma_w, double* scores,
                       int n_planes, int H, int W, int row_stride, size_t plane_stride,
                       size_t sigma_w_plane_stride, float alpha);


/* ==========================================================================
 * Full-frame mode ("tile=None", the reference's own semantics): ONE dense
 * DCT + SVD over the whole plane.  Host-pointer entry points, one plane per
 * call; L = min(H, W).  Singular values are returned sorted descending like
 * np.linalg.svd; singular vectors are unique up to sign.
 * ========================================================================== */

/* Replaces  C = dct2(Y); Uc, Sc, Vct = np.linalg.svd(C); S_[:K] = Sc[:K] +
 * alpha*Sw[:K]; Cw = Uc @ diag(S_) @ Vct; Yw = idct2(Cw); clip/astype(uint8)
 * (single:172-177, 27; per colour plane single:127-147).
 *   sigma_w [L] watermark singular values (sorted), sigma_c [L] out,
 *   yw optional [H][W] float32 unclipped stego, K in 0..L. */
int wm_ref_embed_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego,
                    float* sigma_c, float* yw, int H, int W, int row_stride, float alpha, int K);

/* Replaces  _, S_cw, _ = np.linalg.svd(dct2(Y))  (single:205, 297).  sigma [L]. */
int wm_ref_sigma_u8(wm_ctx* ctx, const uint8_t* plane, float* sigma, int H, int W, int row_stride);

/* Batched forms: n_planes planes (B,G,R of a colour image - single:127-147 - or frames
 * of a video) go through every launch together (the per-plane SVD is latency-bound on
 * its own).  sigma_w: plane p reads sigma_w + p * sigma_w_plane_stride (0 = shared);
 * sigma_c / sigma: [n_planes][L]; yw: [n_planes][H][W]. */
int wm_ref_embed_planes_u8(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego,
                           float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                           size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K);
int wm_ref_sigma_planes_u8(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H, int W,
                           int row_stride, size_t plane_stride);
/* The same embed for a caller that is still computing sigma_w: single:172-173 are two independent decompositions (the host
 * plane's and the watermark's), and one full-frame SVD occupies a fraction of the chip - run on two contexts and two host
 * threads they overlap.  sigma_w is first read after the host planes' decomposition; until then *sigma_w_ready may be 0.
 * The caller stores > 0 (release order) once sigma_w is written, or < 0 to abandon the call (returns WM_ERR_BADARG).
 * sigma_w_ready == NULL: sigma_w is valid on entry (wm_ref_embed_planes_u8). */
int wm_ref_embed_planes_u8_when(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, const int* sigma_w_ready,
                                uint8_t* stego, float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                                size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K);

/* Replaces  Wm = dct2(wy_s); Uw, Sw, Vwt = np.linalg.svd(Wm, full_matrices=False)
 * (single:173, 131-134) when apply_dct != 0 (plain thin SVD of the plane otherwise).
 *   U [H][L], S [L], Vt [L][W]. */
int wm_ref_svd_f32(wm_ctx* ctx, const float* plane, float* U, float* S, float* Vt, int H, int W,
                   int row_stride, int apply_dct);

/* The same for n_planes planes in ONE batch (the B, G, R planes of a colour watermark, single:128-134:
 * `UWb, SWb, VWbt = svd(dct2(wb_s))` ... three times): every launch carries all planes.
 *   planes [n][H][row_stride], U [n][H][L], S [n][L], Vt [n][L][W], host memory. */
int wm_ref_svd_planes_f32(wm_ctx* ctx, const float* planes, float* U, float* S, float* Vt, int n_planes, int H, int W,
                          int row_stride, size_t plane_stride, int apply_dct);

/* Replaces single:205-218: sigma of the stego -> Sw_hat = (S_cw - Sc)/max(alpha,1e-8),
 * Sw_hat[K:] = 0 -> Uw[:L,:L] @ diag(Sw_hat) @ Vwt[:L,:L] zero-padded into H x W
 * (the reference's [:L,:L] truncation on non-square planes is reproduced) -> idct2.
 *   Uw [H][L], Vwt [L][W], out [H][W] float32. */
int wm_ref_extract_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                      const float* Vwt, float* out, int H, int W, int row_stride, float alpha, int K);

/* Replaces single:214-218 on their own, with the estimates given by the caller:
 * Uw[:L,:L] @ diag(sw_hat[:L]) @ Vwt[:L,:L] into the top-left corner of a zero H x W plane
 * (single:215-217), then idct2 (single:218).  L <= min(H, W) is the reference's truncation
 * length `min(len(Sc), len(S_cw), Uw.shape[0], Vwt.shape[0])` (single:210): it is shorter than
 * the meta's own when the stego handed to extract is not the size the meta was written for
 * (resized / cropped stego) - the drop-in then takes sigma of the stego with wm_ref_sigma_u8,
 * forms Sw_hat on the host (single:211-213) and calls this.
 *   Uw [H][min(H,W)], sw_hat [L], Vwt [min(H,W)][W], out [H][W] float32, all host memory. */
int wm_ref_reconstruct_f32(wm_ctx* ctx, const float* Uw, const float* sw_hat, const float* Vwt, float* out,
                           int H, int W, int L);

/* The same for n_planes stego planes that share ONE watermark decomposition (the frames
 * of a clip, single:205-218 applied per frame): their SVDs run as one batch.
 *   sigma_c [n_planes][L], out [n_planes][H][W] float32. */
int wm_ref_extract_planes_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                             const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                             size_t plane_stride, float alpha, int K);

/* Device-pointer forms of the batched full-frame entry points: the planes and the big factors
 * (Uw [H][L], Vwt [L][W], out, yw) are DEVICE memory and never cross PCIe - the frames of a clip
 * or the per-rank frame range of bench.py --mode fullframe stay resident.  The meta-sized vectors
 * (sigma_w, sigma_c, sigma, scores) are device memory too; they are sorted / classified on the
 * host inside the call (L floats per plane), so these calls synchronise the context's stream.
 * Same statements of the reference as the host-pointer forms above. */
int wm_ref_embed_planes_u8_dev(wm_ctx* ctx, const uint8_t* host, const float* sigma_w, uint8_t* stego,
                               float* sigma_c, float* yw, int n_planes, int H, int W, int row_stride,
                               size_t plane_stride, size_t sigma_w_plane_stride, float alpha, int K);
int wm_ref_sigma_planes_u8_dev(wm_ctx* ctx, const uint8_t* planes, float* sigma, int n_planes, int H, int W,
                               int row_stride, size_t plane_stride);
int wm_ref_extract_planes_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw,
                                 const float* Vwt, float* out, int n_planes, int H, int W, int row_stride,
                                 size_t plane_stride, float alpha, int K);
int wm_ref_detect_planes_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                                double* scores, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                                float alpha);

/* Diagnostics: outer Jacobi sweeps the last full-frame SVD on this context needed. */
int wm_ref_last_sweeps(wm_ctx* ctx, int* sweeps_out);
/* Diagnostics (roofline accounting of bench.py): matrix-core flops the block Jacobi of the last full-frame SVD on this
 * context issued - Gram and rotation products as launched, pairs skipped as converged included - and whether it ran the
 * two-level (super-block) scheme (1) or the flat tournament (0). */
int wm_ref_last_flops(wm_ctx* ctx, double* flops_out, int* two_level_out);

/* Replaces single:297-301 + _nc (single:284-289): score over all L singular values. */
int wm_ref_detect_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                     double* score, int H, int W, int row_stride, float alpha);

/* The same for n_planes stego planes carrying ONE watermark (frames of a clip): their SVDs
 * run as one batch.  sigma_c [n_planes][L], sigma_w [L], scores [n_planes]. */
int wm_ref_detect_planes_u8(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* sigma_w,
                            double* scores, int n_planes, int H, int W, int row_stride, size_t plane_stride,
                            float alpha);

/* ==========================================================================
 * Pixel-side kernels either side of the hot path.  Colour
 * conversion is OpenCV's 8-bit fixed point, bit-exact integer work; images
 * are contiguous interleaved 3-channel uint8 (n_px pixels), planes contiguous.
 * *_dev: device pointers (16-byte aligned), asynchronous on the context stream.
 * ========================================================================== */
/* cv2.cvtColor(BGR2YCrCb) / (YCrCb2BGR) / (BGR2GRAY)          single:22, 30, 45, 170 */
int wm_bgr_to_ycrcb_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* ycrcb, size_t n_px);
int wm_ycrcb_to_bgr_u8_dev(wm_ctx* ctx, const uint8_t* ycrcb, uint8_t* bgr, size_t n_px);
int wm_bgr_to_gray_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* gray, size_t n_px);
/* _to_Y (single:21-24): the Y plane of BGR2YCrCb only */
int wm_bgr_to_y_u8_dev(wm_ctx* ctx, const uint8_t* bgr, uint8_t* y, size_t n_px);
/* _from_Y (single:26-30): YCrCb of `bgr` with Y replaced by y_new, back to BGR */
int wm_replace_y_u8_dev(wm_ctx* ctx, const uint8_t* bgr, const uint8_t* y_new, uint8_t* bgr_out, size_t n_px);
/* sum of squared differences of two uint8 buffers (exact integer; psnr of single:38-42 follows) */
int wm_sqdiff_u8_dev(wm_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, unsigned long long* ssd_dev);
/* mean SSIM (single:44-57: 11x11 sigma 1.5 Gaussian, reflect-101 border) of two planes;
 * strides in elements; kind bit0 / bit1: img1 / img2 is float32 instead of uint8 */
int wm_ssim_dev(wm_ctx* ctx, const void* img1, size_t stride1, const void* img2, size_t stride2,
                int H, int W, int kind, double* ssim_dev);
/* uint8(clip(cv2.normalize(x, 0, 255, NORM_MINMAX), 0, 255))  (single:221-222); do_norm=0: clip only.
 * x must be 16-byte aligned (any wm_malloc'd plane is). */
int wm_normalize_u8_dev(wm_ctx* ctx, const float* x, size_t n, int do_norm, uint8_t* out);
/* Keyed scramble / unscramble of the watermark plane, single:66-80 (`_permute`: flat[idx];
 * `_unpermute`: inv[idx] = arange; flat[inv]).  idx is the int32 copy of the permutation NumPy's
 * PCG64 shuffle produced on the host (bit-exact there by construction; this is only the index pass).
 * n elements per plane, n_planes planes share idx; dst float32 [n_planes][n], never in place.
 *   permute:    dst[i] = (float) src[idx[i]]        unpermute:  dst[idx[i]] = src[i] */
int wm_permute_u8_f32_dev(wm_ctx* ctx, const uint8_t* src, const int* idx, float* dst, size_t n, int n_planes);
int wm_permute_f32_dev(wm_ctx* ctx, const float* src, const int* idx, float* dst, size_t n, int n_planes);
int wm_unpermute_f32_dev(wm_ctx* ctx, const float* src, const int* idx, float* dst, size_t n, int n_planes);
/* The end of the reference's extract in one routed, fully coalesced chain (single:74-80 `_unpermute` +
 * single:221-222 cv2.normalize(NORM_MINMAX) / clip / uint8, per plane):
 *     dst[p][idx[i]] = uint8(clip((src[p][i] - min src[p]) * 255 / (max src[p] - min src[p]), 0, 255))
 * (do_norm == 0: clip + truncate only).  A random permutation of a whole plane makes the literal index pass touch one
 * cache line per element; a wm_route factors idx ONCE per key into two block-local permutations around a block
 * transpose, after which the per-frame work streams (csrc/wm_route.hip).  wm_route_create_dev reads the int32 index
 * from device memory, synchronises the stream and returns WM_ERR_BADARG when idx is not a permutation of 0..n-1;
 * the route holds 6 bytes per element of device memory until wm_route_destroy.  src [n_planes][n] float32,
 * dst [n_planes][n] uint8, min / max per plane; bytes identical to wm_unpermute_f32_dev + wm_normalize_u8_dev. */
typedef struct wm_route wm_route;
int wm_route_create_dev(wm_ctx* ctx, const int* idx, size_t n, wm_route** route_out);
int wm_route_destroy(wm_ctx* ctx, wm_route* route);
int wm_unpermute_normalize_u8_dev(wm_ctx* ctx, const float* src, const wm_route* route, uint8_t* dst, size_t n,
                                  int n_planes, int do_norm);
/* The reference's extract per plane in ONE call (single:203-222 gray, :232-274 per colour plane): sigma of every stego tile,
 * (S_cw - Sc) / max(alpha, 1e-8) with [K:] = 0, the rank-8 product with the watermark's factors (px != 0: pixel-domain
 * factors from wm_tile_factors_to_pixel_dev, no per-tile IDCT; px == 0: Uw / Vwt as stored in the meta), the keyed
 * unscramble through `route`, min-max normalise (do_norm) / clip / uint8.  The float estimate never leaves a buffer of
 * the context and its min / max are taken by the extract kernel itself.  out [n_planes][H*W] uint8; bytes identical
 * to wm_extract_tiles[_px]_u8_dev + wm_unpermute_normalize_u8_dev. */
int wm_extract_unscrambled_u8_dev(wm_ctx* ctx, const uint8_t* stego, const float* sigma_c, const float* Uw, const float* Vwt,
                                  const wm_route* route, uint8_t* out, int n_planes, int H, int W, int row_stride,
                                  size_t plane_stride, size_t uv_plane_stride, float alpha, int K, int px, int do_norm);
/* the scramble direction through the same route (single:66-72, 124-126): dst[p][i] = (float) src[p][idx[i]], uint8 planes
 * in, float32 out; values identical to wm_permute_u8_f32_dev */
int wm_permute_u8_f32_routed_dev(wm_ctx* ctx, const uint8_t* src, const wm_route* route, float* dst, size_t n, int n_planes);
/* host-pointer conveniences; op: 0 BGR->YCrCb, 1 YCrCb->BGR, 2 BGR->gray plane, 3 BGR->Y plane,
 * 4 replace Y (plane_in) and return BGR */
int wm_color_u8(wm_ctx* ctx, int op, const uint8_t* in3, const uint8_t* plane_in, uint8_t* out3,
                uint8_t* plane_out, size_t n_px);
int wm_psnr_u8(wm_ctx* ctx, const uint8_t* a, const uint8_t* b, size_t n, double* psnr_out);
int wm_ssim(wm_ctx* ctx, const void* img1, const void* img2, int H, int W, int kind, double* ssim_out);
int wm_normalize_u8(wm_ctx* ctx, const float* x, size_t n, int do_norm, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* WMHIP_H */
