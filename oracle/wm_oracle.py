"""CPU oracle for the DCT-SVD watermark hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path is the HIP library behind ``include/wmhip.h``.

What it restates
----------------
The array arithmetic of the reference's authoritative implementation,
``app_dct_svd_single.py:13-86,112-318`` (citations below are into
``/root/reference``; ``single`` = ``app_dct_svd_single.py``,
``core`` = ``dct_svd_core_secure.py``), statement by statement, with

* ``cv2.dct / cv2.idct``   -> ``scipy.fft.dctn / idctn(type=2, norm='ortho')``
  (OpenCV's documented definition of the orthonormal DCT-II; opencv-python
  4.12.0.88 is the pin in the reference's venv and is NOT installed here);
* ``np.linalg.svd``        -> the same NumPy call (numpy 2.2.6 here as in the
  reference's venv): float32 input is up-cast, solved by LAPACK ``dgesdd`` in
  float64, results cast back to float32;
* ``cv2.cvtColor / resize / GaussianBlur / normalize`` -> NumPy/SciPy
  restatements of OpenCV's documented fixed-point formulas (host glue only;
  the BASELINE configs feed Y / B,G,R planes directly).

Two modes, one code path:

* ``tile=None``  reference semantics: one DCT + one SVD over the whole plane.
* ``tile=8``     the north_star's kernel formulation: the *same per-matrix
  arithmetic* applied to every 8x8 tile independently (what the HIP kernels
  compute).  Border rows/columns that do not fill a tile are passed through
  unchanged by embed and read as zero by extract.

PARITY UNPINNED
---------------
The reference ships no tests, golden vectors, sample images or known-answer
values, and cannot run here (``import cv2`` -> ModuleNotFoundError; its
vendored venv is win_amd64 with the binaries missing).  Nothing outside this
file therefore pins its outputs: the golden fixtures under ``tests/golden``
are ORACLE-generated, not reference-generated.  What *is* pinned
independently: the DCT against the closed-form basis ``D X D^T``, the SVD by
reconstruction/orthogonality, the key/permutation/HMAC glue against
hashlib/NumPy known answers (tests/test_oracle.py).
"""
from __future__ import annotations

import hashlib
import hmac as _hmac
from typing import Optional

import numpy as np
import scipy.fft
import scipy.ndimage

K_FRAC_DEFAULT = 0.6  # single:13  "embed top 60% singular values"
TILE = 8
K_FLOOR_DEFAULT = 8   # the literal 8 in  K = max(8, int(kfrac * L))   single:136,174


# --------------------------------------------------------------------------
# a2 / a6 : orthonormal 2-D DCT-II and its inverse          single:32-36
# --------------------------------------------------------------------------
def dct2(x: np.ndarray) -> np.ndarray:
    """``cv2.dct(x.astype(np.float32))``  (single:32-33, core:31-32)."""
    return scipy.fft.dctn(x.astype(np.float32), type=2, norm="ortho").astype(np.float32)


def idct2(X: np.ndarray) -> np.ndarray:
    """``cv2.idct(X.astype(np.float32))``  (single:35-36, core:34-35)."""
    return scipy.fft.idctn(X.astype(np.float32), type=2, norm="ortho").astype(np.float32)


def dct_basis(n: int) -> np.ndarray:
    """Closed-form orthonormal DCT-II basis D (float64): C = D @ X @ D.T.

    D[k, m] = sqrt(a_k / n) * cos(pi * (2m + 1) * k / (2n)), a_0 = 1, a_k = 2.
    Used only to pin ``dct2``.
    """
    k = np.arange(n)[:, None].astype(np.float64)
    m = np.arange(n)[None, :].astype(np.float64)
    D = np.cos(np.pi * (2 * m + 1) * k / (2 * n)) * np.sqrt(2.0 / n)
    D[0, :] = np.sqrt(1.0 / n)
    return D


# --------------------------------------------------------------------------
# tiling helpers (tile-mode only; nothing in the reference)
# --------------------------------------------------------------------------
def tile_grid(H: int, W: int, tile: int = TILE):
    return H // tile, W // tile


def to_tiles(plane: np.ndarray, tile: int = TILE) -> np.ndarray:
    """[H, W] -> [nby, nbx, tile, tile] over the region the tiles cover."""
    H, W = plane.shape
    nby, nbx = tile_grid(H, W, tile)
    body = plane[: nby * tile, : nbx * tile]
    return body.reshape(nby, tile, nbx, tile).transpose(0, 2, 1, 3)


def from_tiles(tiles: np.ndarray, out: np.ndarray) -> np.ndarray:
    """Write [nby, nbx, t, t] back into the tile-covered region of ``out``."""
    nby, nbx, t, _ = tiles.shape
    out[: nby * t, : nbx * t] = tiles.transpose(0, 2, 1, 3).reshape(nby * t, nbx * t)
    return out


def _dct_tiles(t: np.ndarray) -> np.ndarray:
    return scipy.fft.dctn(t.astype(np.float32), type=2, norm="ortho", axes=(-2, -1)).astype(np.float32)


def _idct_tiles(t: np.ndarray) -> np.ndarray:
    return scipy.fft.idctn(t.astype(np.float32), type=2, norm="ortho", axes=(-2, -1)).astype(np.float32)


def k_of(L: int, kfrac: float, k_floor: int = K_FLOOR_DEFAULT) -> int:
    """``K = max(8, int(kfrac * L))``  (single:136,174,211,252).

    ``k_floor`` exposes the literal 8: at tile=8 the formula is 8 for every
    kfrac <= 1, so a mid-band sweep needs k_floor < 8.
    """
    return max(int(k_floor), int(kfrac * L))


# --------------------------------------------------------------------------
# a3 : thin SVD                                  single:128-134,172-173,205
# --------------------------------------------------------------------------
def svd_f32(C: np.ndarray):
    """``np.linalg.svd(C, full_matrices=False)`` on float32 (batched OK)."""
    return np.linalg.svd(C.astype(np.float32), full_matrices=False)


def sigma_f32(C: np.ndarray) -> np.ndarray:
    """Singular values the way extract/detect obtain them: the reference runs
    the *full* SVD and discards U,V (``_, S_cw, _ =``  single:205,297)."""
    return np.linalg.svd(C.astype(np.float32), full_matrices=False)[1]


# --------------------------------------------------------------------------
# a1..a7 : embed one plane                        single:168-177 (gray branch)
#                                                 single:127-147 (per colour plane)
# --------------------------------------------------------------------------
def watermark_decompose(wy_s: np.ndarray, tile: Optional[int] = None):
    """``Wm = dct2(wy_s); Uw, Sw, Vwt = svd(Wm)``  (single:173, 131-134)."""
    if tile is None:
        return svd_f32(dct2(wy_s))
    return svd_f32(_dct_tiles(to_tiles(wy_s.astype(np.float32), tile)))


def embed_plane(Y: np.ndarray, wy_s: np.ndarray, alpha: float,
                kfrac: float = K_FRAC_DEFAULT, tile: Optional[int] = None,
                k_floor: int = K_FLOOR_DEFAULT, wm_svd=None) -> dict:
    """Embed the scrambled watermark plane ``wy_s`` into the host plane ``Y``.

    Y, wy_s: float32 [H, W] (Y holds uint8-valued samples, single:24,122).
    Returns stego (uint8, clip + truncation  single:27,145-147), Yw (float32,
    unclipped - what gray-mode SSIM sees, single:190), Sc, Sw, Uw, Vwt.
    """
    Y = Y.astype(np.float32)
    H, W = Y.shape
    if wm_svd is None:
        wm_svd = watermark_decompose(wy_s, tile)
    Uw, Sw, Vwt = wm_svd
    if tile is None:
        C = dct2(Y)                                              # single:172
        Uc, Sc, Vct = svd_f32(C)                                 # single:172
        L = min(len(Sc), len(Sw)); K = k_of(L, kfrac, k_floor)   # single:174
        S_ = Sc.copy(); S_[:K] = Sc[:K] + alpha * Sw[:K]         # single:175
        Cw = (Uc @ np.diag(S_) @ Vct).astype(np.float32)         # single:176
        Yw = idct2(Cw)                                           # single:177
    else:
        T = to_tiles(Y, tile)
        C = _dct_tiles(T)
        Uc, Sc, Vct = svd_f32(C)                                 # [nby,nbx,8,8],[..,8],[..,8,8]
        L = min(Sc.shape[-1], Sw.shape[-1]); K = k_of(L, kfrac, k_floor)
        S_ = Sc.copy(); S_[..., :K] = Sc[..., :K] + alpha * Sw[..., :K]
        # U @ diag(S_) == U * S_ column-wise (the zero products add exactly)
        Cw = np.matmul(Uc * S_[..., None, :], Vct).astype(np.float32)
        Yw = Y.copy()
        from_tiles(_idct_tiles(Cw), Yw)                          # ragged border: pass-through
    stego = np.clip(Yw, 0, 255).astype(np.uint8)                 # single:27  (truncation)
    return dict(stego=stego, Yw=Yw, Sc=Sc, Sw=Sw, Uw=Uw, Vwt=Vwt, K=K)


# --------------------------------------------------------------------------
# a8 / a9 : extract one plane                     single:204-218, 248-264
# --------------------------------------------------------------------------
def stego_sigma(Y: np.ndarray, tile: Optional[int] = None) -> np.ndarray:
    """``Cw = dct2(Y); _, S_cw, _ = svd(Cw)``  (single:205,234-236,297)."""
    Y = Y.astype(np.float32)
    if tile is None:
        return sigma_f32(dct2(Y))
    return sigma_f32(_dct_tiles(to_tiles(Y, tile)))


def extract_plane(Y: np.ndarray, Sc: np.ndarray, Uw: np.ndarray, Vwt: np.ndarray,
                  alpha: float, kfrac: float, H: int, W: int,
                  tile: Optional[int] = None, k_floor: int = K_FLOOR_DEFAULT) -> np.ndarray:
    """Scrambled watermark estimate ``wy_s`` (float32 [H, W]) from a stego plane.

    Reproduces the ``[:L, :L]`` truncation quirk on non-square planes
    (single:214-217).
    """
    S_cw = stego_sigma(Y, tile)
    a = max(alpha, 1e-8)                                          # single:212
    if tile is None:
        L = min(len(Sc), len(S_cw), Uw.shape[0], Vwt.shape[0])    # single:210
        K = k_of(L, kfrac, k_floor)                               # single:211
        Sw_hat = (S_cw[:L] - Sc[:L]) / a                          # single:212
        Sw_hat[K:] = 0                                            # single:213
        Wm_hat = (Uw[:L, :L] @ np.diag(Sw_hat) @ Vwt[:L, :L]).astype(np.float32)  # single:214
        Wm_full = np.zeros((H, W), np.float32)                    # single:215
        hh = min(Wm_hat.shape[0], H); ww = min(Wm_hat.shape[1], W)
        Wm_full[:hh, :ww] = Wm_hat[:hh, :ww]                      # single:216-217
        return idct2(Wm_full)                                     # single:218
    L = min(Sc.shape[-1], S_cw.shape[-1], Uw.shape[-2], Vwt.shape[-2])
    K = k_of(L, kfrac, k_floor)
    Sw_hat = ((S_cw[..., :L] - Sc[..., :L]) / a).astype(np.float32)
    Sw_hat[..., K:] = 0
    Wm_hat = np.matmul(Uw[..., :L, :L] * Sw_hat[..., None, :], Vwt[..., :L, :L]).astype(np.float32)
    out = np.zeros((H, W), np.float32)
    from_tiles(_idct_tiles(Wm_hat), out)
    return out


# --------------------------------------------------------------------------
# a10 : detect                                   single:284-289, 295-318
# --------------------------------------------------------------------------
def nc(a: np.ndarray, b: np.ndarray) -> float:
    """``_nc``  (single:284-289): mean-removed normalised correlation."""
    a = a.astype(np.float32); b = b.astype(np.float32)
    if a.size == 0 or b.size == 0:
        return 0.0
    a = a - np.mean(a); b = b - np.mean(b)
    den = np.linalg.norm(a) * np.linalg.norm(b) + 1e-8
    return float(np.dot(a, b) / den)


def detect_plane(Y: np.ndarray, Sc: np.ndarray, Sw: np.ndarray, alpha: float,
                 tile: Optional[int] = None) -> float:
    """NC between stored Sw and (S_cw - Sc)/alpha over ALL L (no K cut)
    (single:299-301).  tile-mode: over the flattened [nb*8] vectors."""
    S_cw = stego_sigma(Y, tile)
    Sc = Sc.reshape(-1); Sw = Sw.reshape(-1); S_cw = S_cw.reshape(-1)
    L = min(len(Sc), len(S_cw), len(Sw))
    Sw_hat = (S_cw[:L] - Sc[:L]) / max(alpha, 1e-8)
    return nc(Sw[:L], Sw_hat)


# --------------------------------------------------------------------------
# L2 security wrapper                              single:59-86
# --------------------------------------------------------------------------
def derive_key(password: str, nonce: bytes) -> bytes:
    return hashlib.sha256(password.encode("utf-8") + nonce).digest()      # single:59-60


def rng_from_key(key: bytes) -> np.random.Generator:
    seed = int.from_bytes(key[:8], "big", signed=False)                   # single:63
    return np.random.default_rng(seed)                                    # single:64


def permutation(H: int, W: int, rng: np.random.Generator) -> np.ndarray:
    idx = np.arange(H * W); rng.shuffle(idx)                              # single:68-69,124
    return idx


def permute(img: np.ndarray, idx: np.ndarray) -> np.ndarray:
    H, W = img.shape[:2]
    return img.reshape(-1)[idx].reshape(H, W).astype(np.float32)          # single:70-72,125


def unpermute(img_scrambled: np.ndarray, idx: np.ndarray) -> np.ndarray:
    H, W = img_scrambled.shape[:2]
    inv = np.empty_like(idx); inv[idx] = np.arange(idx.size)              # single:77-78
    return img_scrambled.reshape(-1)[inv].reshape(H, W)                   # single:79


def hmac_digest(key: bytes, parts) -> bytes:
    h = _hmac.new(key, b"", hashlib.sha256)                               # single:83
    for p in parts:
        h.update(p)
    return h.digest()


# --------------------------------------------------------------------------
# L4' metrics                                      single:38-57
# --------------------------------------------------------------------------
def psnr(a: np.ndarray, b: np.ndarray) -> float:
    a = a.astype(np.float32); b = b.astype(np.float32)
    mse = float(np.mean((a - b) ** 2))
    if mse <= 1e-12:
        return 99.0
    return float(20.0 * np.log10(255.0 / max(np.sqrt(mse), 1e-12)))


def _gauss_kernel(ksize: int = 11, sigma: float = 1.5) -> np.ndarray:
    # cv2.getGaussianKernel: exp(-(i-(k-1)/2)^2 / (2 sigma^2)), normalised to 1
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def gaussian_blur(img: np.ndarray, ksize: int = 11, sigma: float = 1.5) -> np.ndarray:
    """cv2.GaussianBlur(img, (k,k), s) on float32: separable, BORDER_REFLECT_101
    (scipy 'mirror')."""
    k = _gauss_kernel(ksize, sigma)
    t = scipy.ndimage.correlate1d(img.astype(np.float32), k, axis=0, mode="mirror")
    return scipy.ndimage.correlate1d(t, k, axis=1, mode="mirror").astype(np.float32)


def ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    if img1.ndim == 3: img1 = bgr_to_gray(img1)                  # single:45
    if img2.ndim == 3: img2 = bgr_to_gray(img2)
    img1 = img1.astype(np.float32); img2 = img2.astype(np.float32)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1 = gaussian_blur(img1); mu2 = gaussian_blur(img2)
    mu1_sq = mu1 * mu1; mu2_sq = mu2 * mu2; mu1_mu2 = mu1 * mu2
    sigma1_sq = gaussian_blur(img1 * img1) - mu1_sq
    sigma2_sq = gaussian_blur(img2 * img2) - mu2_sq
    sigma12 = gaussian_blur(img1 * img2) - mu1_mu2
    num = (2 * mu1_mu2 + C1) * (2 * sigma12 + C2)
    den = (mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2) + 1e-12
    return float(np.mean(num / den))


# --------------------------------------------------------------------------
# colour / resize glue (OpenCV fixed-point formulas restated; cannot be
# checked against cv2 here - SURVEY.md section 8c)
# --------------------------------------------------------------------------
def bgr_to_gray(bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) on uint8: (B*3735 + G*19235 + R*9798 + 2^14) >> 15."""
    b = bgr[..., 0].astype(np.int64); g = bgr[..., 1].astype(np.int64); r = bgr[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def bgr_to_ycrcb(bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2YCrCb) on uint8, 14-bit fixed point."""
    b = bgr[..., 0].astype(np.int64); g = bgr[..., 1].astype(np.int64); r = bgr[..., 2].astype(np.int64)
    half = 1 << 13; delta = 128 << 14
    y = (r * 4899 + g * 9617 + b * 1868 + half) >> 14
    cr = ((r - y) * 11682 + delta + half) >> 14
    cb = ((b - y) * 9241 + delta + half) >> 14
    return np.stack([np.clip(y, 0, 255), np.clip(cr, 0, 255), np.clip(cb, 0, 255)], -1).astype(np.uint8)


def ycrcb_to_bgr(ycc: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(YCrCb2BGR) on uint8, 14-bit fixed point."""
    y = ycc[..., 0].astype(np.int64); cr = ycc[..., 1].astype(np.int64) - 128; cb = ycc[..., 2].astype(np.int64) - 128
    half = 1 << 13
    b = y + ((cb * 29049 + half) >> 14)
    g = y + ((cb * -5636 + cr * -11698 + half) >> 14)
    r = y + ((cr * 22987 + half) >> 14)
    return np.stack([np.clip(b, 0, 255), np.clip(g, 0, 255), np.clip(r, 0, 255)], -1).astype(np.uint8)


def resize_area(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv2.resize(img, (W, H), INTER_AREA) on uint8.

    Integer up-scale factors reduce to pixel replication and integer
    down-scale factors to box means (round-half-up); other ratios use exact
    fractional box coverage when both axes shrink, and the area-variant of linear
    on BOTH axes as soon as one of them enlarges (cv::resize decides jointly).
    Unpinned against OpenCV itself (cv2 absent): exact weights, round-half-up.
    """
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img.copy()
    src = img.astype(np.float64)

    both_shrink = (h >= H) and (w >= W)

    def axis_weights(n_src, n_dst):
        M = np.zeros((n_dst, n_src), np.float64)
        scale = n_src / n_dst
        if both_shrink:                       # shrinking: fractional box coverage
            for d in range(n_dst):
                lo, hi = d * scale, (d + 1) * scale
                s0 = int(np.floor(lo)); s1 = min(int(np.ceil(hi)), n_src)
                for s in range(s0, s1):
                    M[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
                M[d] /= M[d].sum()
        else:                                 # OpenCV's INTER_AREA-as-linear (either axis enlarges)
            inv = 1.0 / scale
            for d in range(n_dst):
                s = int(np.floor(d * scale))
                fx = (d + 1) - (s + 1) * inv
                fx = 0.0 if fx <= 0 else fx - np.floor(fx)
                s2 = min(s + 1, n_src - 1)
                M[d, s] += 1.0 - fx; M[d, s2] += fx
        return M

    My = axis_weights(h, H); Mx = axis_weights(w, W)
    if src.ndim == 2:
        out = My @ src @ Mx.T
    else:                                     # every channel is resized like a gray image
        out = np.stack([My @ src[..., c] @ Mx.T for c in range(src.shape[2])], axis=-1)
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def normalize_minmax(x: np.ndarray) -> np.ndarray:
    """cv2.normalize(x, None, 0, 255, NORM_MINMAX) on float32 (single:221)."""
    x = x.astype(np.float32)
    lo = float(x.min()); hi = float(x.max())
    scale = (255.0 / (hi - lo)) if (hi - lo) > np.finfo(np.float64).eps else 0.0
    return ((x - np.float32(lo)) * np.float32(scale)).astype(np.float32)


# --------------------------------------------------------------------------
# L4 array-level pipelines (embed/extract/detect without file I/O)
# --------------------------------------------------------------------------
def embed_arrays(cover_bgr: np.ndarray, wm_bgr: np.ndarray, password: str, nonce: bytes,
                 alpha: float = 0.1, color: bool = False, kfrac: float = K_FRAC_DEFAULT,
                 tile: Optional[int] = None, k_floor: int = K_FLOOR_DEFAULT) -> dict:
    """``embed`` (single:112-190) from decoded arrays, with the nonce injected
    (the reference draws ``os.urandom(8)``, single:119)."""
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để nhúng.")                # single:115-116
    H, W = cover_bgr.shape[:2]
    wm = resize_area(wm_bgr, W, H)                                          # single:118
    key = derive_key(password, nonce); rng = rng_from_key(key)              # single:119
    if color:
        idx = permutation(H, W, rng)                                        # single:124
        planes = []
        for ch in range(3):                                                 # b, g, r  single:122-147
            host = cover_bgr[..., ch].astype(np.float32)
            w_s = permute(wm[..., ch].astype(np.float32), idx)
            planes.append(embed_plane(host, w_s, alpha, kfrac, tile, k_floor))
        stego = np.stack([p["stego"] for p in planes], -1)                  # single:145-147
        names = ("b", "g", "r")
        meta = dict(mode="color", payload_type="image",
                    shape=np.array((H, W)), alpha=float(alpha), kfrac=float(kfrac),
                    nonce=np.frombuffer(nonce, dtype=np.uint8))
        for n, p in zip(names, planes):
            meta["S" + n] = p["Sc"]; meta["UW" + n] = p["Uw"]
            meta["VW" + n + "t"] = p["Vwt"]; meta["SW" + n] = p["Sw"]
        digest = hmac_digest(key, [meta["Sb"].tobytes(), meta["Sg"].tobytes(), meta["Sr"].tobytes(),
                                   meta["UWb"].tobytes(), meta["UWg"].tobytes(), meta["UWr"].tobytes(),
                                   meta["VWbt"].tobytes(), meta["VWgt"].tobytes(), meta["VWrt"].tobytes()])
        meta["digest"] = np.frombuffer(digest, dtype=np.uint8)              # single:152-156
        return dict(stego=stego, meta=meta, psnr=psnr(cover_bgr, stego),
                    ssim=ssim(cover_bgr, stego), planes=planes)             # single:167
    ycc = bgr_to_ycrcb(cover_bgr)                                           # single:169 (_to_Y)
    Y = ycc[..., 0].astype(np.float32)
    wy = bgr_to_gray(wm).astype(np.float32)                                 # single:170
    idx = permutation(H, W, rng)
    wy_s = permute(wy, idx)                                                 # single:171
    p = embed_plane(Y, wy_s, alpha, kfrac, tile, k_floor)
    out = ycc.copy(); out[..., 0] = p["stego"]                              # single:27-29 (_from_Y)
    stego = ycrcb_to_bgr(out)                                               # single:30
    digest = hmac_digest(key, [p["Sc"].tobytes(), p["Uw"].tobytes(), p["Vwt"].tobytes()])  # single:182
    meta = dict(mode="gray", payload_type="image", Sc=p["Sc"], Uw=p["Uw"], Vwt=p["Vwt"], Sw=p["Sw"],
                shape=np.array((H, W)), alpha=float(alpha), kfrac=float(kfrac),
                nonce=np.frombuffer(nonce, dtype=np.uint8),
                digest=np.frombuffer(digest, dtype=np.uint8))               # single:183-189
    return dict(stego=stego, meta=meta, psnr=psnr(cover_bgr, stego),
                ssim=ssim(bgr_to_gray(cover_bgr), p["Yw"]), planes=[p])     # single:190


def extract_arrays(stego_bgr: np.ndarray, meta: dict, password: str, normalize: bool = True,
                   tile: Optional[int] = None, k_floor: int = K_FLOOR_DEFAULT) -> np.ndarray:
    """``extract`` (single:192-282) up to - not including - NL-means denoise and
    CLAHE/unsharp (cosmetic, OpenCV-only, wrapped in try/except in the
    reference: single:223-227,275-277).  Returns uint8 [H,W] or [H,W,3]."""
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để giải trích.")          # single:193-194
    mode = str(meta["mode"]); alpha = float(meta["alpha"])
    H, W = map(int, meta["shape"])
    nonce = bytes(bytearray(np.asarray(meta["nonce"]).astype(np.uint8).tolist()))
    digest = bytes(bytearray(np.asarray(meta["digest"]).astype(np.uint8).tolist()))
    key = derive_key(password, nonce)
    kfrac = float(meta.get("kfrac", K_FRAC_DEFAULT))
    if mode == "gray":
        Sc, Uw, Vwt = meta["Sc"], meta["Uw"], meta["Vwt"]
        expected = hmac_digest(key, [Sc.tobytes(), Uw.tobytes(), Vwt.tobytes()])
        if not _hmac.compare_digest(expected, digest):
            raise ValueError("Sai mật khẩu hoặc meta không khớp.")          # single:208-209
        Y = bgr_to_ycrcb(stego_bgr)[..., 0].astype(np.float32)             # single:204
        wy_s = extract_plane(Y, Sc, Uw, Vwt, alpha, kfrac, H, W, tile, k_floor)
        idx = permutation(H, W, rng_from_key(key))                          # single:219
        wy = unpermute(wy_s, idx)                                           # single:220
        if normalize:
            wy = normalize_minmax(wy)                                       # single:221
        return np.clip(wy, 0, 255).astype(np.uint8)                         # single:222
    names = ("b", "g", "r")
    parts = [meta["S" + n].tobytes() for n in names] + [meta["UW" + n].tobytes() for n in names] \
        + [meta["VW" + n + "t"].tobytes() for n in names]
    if not _hmac.compare_digest(hmac_digest(key, parts), digest):
        raise ValueError("Sai mật khẩu hoặc meta không khớp.")              # single:246-247
    idx = permutation(H, W, rng_from_key(key))                              # single:265
    outs = []
    for ch, n in enumerate(names):
        Yc = stego_bgr[..., ch].astype(np.float32)                          # single:232
        w_s = extract_plane(Yc, meta["S" + n], meta["UW" + n], meta["VW" + n + "t"],
                            alpha, kfrac, H, W, tile, k_floor)
        w = unpermute(w_s, idx)
        if normalize:
            w = normalize_minmax(w)                                         # single:269-271
        outs.append(np.clip(w, 0, 255).astype(np.uint8))                    # single:272-274
    return np.stack(outs, -1)


def detect_arrays(stego_bgr: np.ndarray, meta: dict, thresh: float = 0.6,
                  tile: Optional[int] = None):
    """``detect`` (single:291-318)."""
    mode = str(meta["mode"]); alpha = float(meta["alpha"])
    if mode == "gray":
        Y = bgr_to_ycrcb(stego_bgr)[..., 0].astype(np.float32)
        score = detect_plane(Y, meta["Sc"], meta["Sw"], alpha, tile)
        return bool(score >= thresh), float(score)
    scores = []
    for ch, n in enumerate(("b", "g", "r")):
        scores.append(detect_plane(stego_bgr[..., ch].astype(np.float32),
                                   meta["S" + n], meta["SW" + n], alpha, tile))
    score = (scores[0] + scores[1] + scores[2]) / 3.0                       # single:317
    return bool(score >= thresh), float(score)
