"""ctypes binding of libwmhip.so (include/wmhip.h) - the thin Python side of
the C ABI.  No CPU fallback: if the HIP library is missing or there is no GPU,
calls raise (``WmLibraryError`` / ``WmError``) instead of computing anything
on the host.

The error mapping mirrors what the reference's callers see
(app_dct_svd_single.py:115-116,208-209, numpy LinAlgError):
  WM_ERR_BADARG -> ValueError, WM_ERR_NOCONV -> numpy.linalg.LinAlgError,
  WM_ERR_NOMEM  -> MemoryError, WM_ERR_HIP -> WmError (RuntimeError).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
from typing import Optional

import numpy as np

WM_OK, WM_ERR_BADARG, WM_ERR_HIP, WM_ERR_NOCONV, WM_ERR_NOMEM = 0, 1, 2, 3, 4
TILE = 8
ABI_VERSION = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# WMHIP_LIB selects another build of the same HIP library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("WMHIP_LIB") or os.path.join(_HERE, "csrc", "libwmhip.so")


class WmLibraryError(ImportError):
    """libwmhip.so is not built / not loadable."""


class WmError(RuntimeError):
    """A HIP runtime call inside libwmhip.so failed."""


_lib = None

_sz = C.c_size_t
_vp = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> argtypes  (every symbol include/wmhip.h declares)
SIGNATURES = {
    "wm_abi_version": [],
    "wm_last_error": [],
    "wm_device_count": [C.POINTER(_i)],
    "wm_create": [_i, _vp, C.POINTER(_vp)],
    "wm_destroy": [_vp],
    "wm_sync": [_vp],
    "wm_check_status": [_vp],
    "wm_malloc": [_vp, _sz, C.POINTER(_vp)],
    "wm_free": [_vp, _vp],
    "wm_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "wm_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "wm_memset": [_vp, _vp, _i, _sz],
    "wm_copy_mapped_dev": [_vp, _vp, _vp, _sz, _i],
    "wm_event_record": [_vp, _i],
    "wm_event_elapsed_ms": [_vp, _i, _i, C.POINTER(_f)],
    "wm_embed_tiles_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_embed_tiles_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_sigma_tiles_u8_dev": [_vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_sigma_tiles_u8": [_vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_svd_tiles_f32_dev": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_svd_tiles_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_extract_tiles_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_extract_tiles_px_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_tile_factors_to_pixel_dev": [_vp, _vp, _vp, _vp, _vp, _sz],
    "wm_extract_tiles_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_extract_tiles_sum_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_reconstruct_tiles_dev": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "wm_reconstruct_tiles": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "wm_detect_tiles_u8_dev": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f],
    "wm_detect_tiles_u8": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f],
    "wm_ref_embed_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i],
    "wm_ref_sigma_u8": [_vp, _vp, _vp, _i, _i, _i],
    "wm_ref_last_sweeps": [_vp, C.POINTER(_i)],
    "wm_ref_last_flops": [_vp, C.POINTER(C.c_double), C.POINTER(_i)],
    "wm_ref_embed_planes_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_ref_embed_planes_u8_when": [_vp, _vp, _vp, C.POINTER(_i), _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_ref_sigma_planes_u8": [_vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_ref_embed_planes_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i],
    "wm_ref_sigma_planes_u8_dev": [_vp, _vp, _vp, _i, _i, _i, _i, _sz],
    "wm_ref_extract_planes_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _f, _i],
    "wm_ref_detect_planes_u8_dev": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _f],
    "wm_ref_svd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "wm_ref_svd_planes_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _i],
    "wm_ref_extract_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i],
    "wm_ref_extract_planes_u8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _f, _i],
    "wm_ref_reconstruct_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "wm_ref_detect_u8": [_vp, _vp, _vp, _vp, C.POINTER(C.c_double), _i, _i, _i, _f],
    "wm_ref_detect_planes_u8": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _f],
    "wm_bgr_to_ycrcb_u8_dev": [_vp, _vp, _vp, _sz],
    "wm_ycrcb_to_bgr_u8_dev": [_vp, _vp, _vp, _sz],
    "wm_bgr_to_gray_u8_dev": [_vp, _vp, _vp, _sz],
    "wm_bgr_to_y_u8_dev": [_vp, _vp, _vp, _sz],
    "wm_replace_y_u8_dev": [_vp, _vp, _vp, _vp, _sz],
    "wm_sqdiff_u8_dev": [_vp, _vp, _vp, _sz, _vp],
    "wm_ssim_dev": [_vp, _vp, _sz, _vp, _sz, _i, _i, _i, _vp],
    "wm_normalize_u8_dev": [_vp, _vp, _sz, _i, _vp],
    "wm_permute_u8_f32_dev": [_vp, _vp, _vp, _vp, _sz, _i],
    "wm_permute_f32_dev": [_vp, _vp, _vp, _vp, _sz, _i],
    "wm_unpermute_f32_dev": [_vp, _vp, _vp, _vp, _sz, _i],
    "wm_route_create_dev": [_vp, _vp, _sz, C.POINTER(_vp)],
    "wm_route_destroy": [_vp, _vp],
    "wm_unpermute_normalize_u8_dev": [_vp, _vp, _vp, _vp, _sz, _i, _i],
    "wm_permute_u8_f32_routed_dev": [_vp, _vp, _vp, _vp, _sz, _i],
    "wm_extract_unscrambled_u8_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _sz, _f, _i, _i, _i],
    "wm_color_u8": [_vp, _i, _vp, _vp, _vp, _vp, _sz],
    "wm_psnr_u8": [_vp, _vp, _vp, _sz, C.POINTER(C.c_double)],
    "wm_ssim": [_vp, _vp, _vp, _i, _i, _i, C.POINTER(C.c_double)],
    "wm_normalize_u8": [_vp, _vp, _sz, _i, _vp],
}


def load_library(path: Optional[str] = None):
    """dlopen libwmhip.so and type every entry point.  Raises WmLibraryError -
    never falls back to a host implementation."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise WmLibraryError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # missing libamdhip64 etc.
        raise WmLibraryError(f"cannot load {p}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the ABI drifted
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "wm_last_error" else _i
    if lib.wm_abi_version() != ABI_VERSION:
        raise WmLibraryError(f"ABI version mismatch: library {lib.wm_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def _raise(lib, rc: int):
    msg = (lib.wm_last_error() or b"").decode("utf-8", "replace")
    if rc == WM_ERR_BADARG:
        raise ValueError(msg)
    if rc == WM_ERR_NOCONV:
        raise np.linalg.LinAlgError(msg)
    if rc == WM_ERR_NOMEM:
        raise MemoryError(msg)
    raise WmError(f"[{rc}] {msg}")


def device_count() -> int:
    lib = load_library()
    n = _i(0)
    rc = lib.wm_device_count(C.byref(n))
    if rc:
        _raise(lib, rc)
    return n.value


def _plane_layout(a: np.ndarray):
    """(n_planes, H, W, row_stride, plane_stride) in elements of a [H,W] or
    [N,H,W] array whose last axis is contiguous."""
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("planes must be [H, W] or [n_planes, H, W]")
    n, H, W = a.shape
    it = a.itemsize
    if W > 1 and a.strides[2] != it:
        raise ValueError("last axis must be contiguous")
    if any(st < 0 or st % it for st in a.strides):
        raise ValueError("negative or unaligned strides are not supported")
    rs = max(W, a.strides[1] // it) if H > 1 else W       # elements between rows
    ps = a.strides[0] // it if n > 1 else rs * H           # elements between planes
    if n > 1 and ps < rs * (H - 1) + W:
        raise ValueError("planes overlap")
    return n, H, W, rs, ps


class Context:
    """One device + one HIP stream (wm_ctx).  ``stream`` may be a raw
    hipStream_t handle (e.g. ``torch.cuda.current_stream().cuda_stream``)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.wm_create(device, _vp(stream) if stream else None, C.byref(h))
        if rc:
            _raise(self.lib, rc)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            for ent in self.__dict__.pop("_idx_cache", []):
                self._drop_index(ent)
            self.lib.wm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(self._h, *args)
        if rc:
            _raise(self.lib, rc)

    # ---- plumbing -------------------------------------------------------
    def sync(self):
        self._call("wm_sync")

    def check_status(self):
        self._call("wm_check_status")

    def malloc(self, nbytes: int) -> int:
        p = _vp()
        self._call("wm_malloc", nbytes, C.byref(p))
        return p.value

    def free(self, dptr: int):
        self._call("wm_free", _vp(dptr))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._call("wm_memcpy_h2d", _vp(dptr), _vp(arr.ctypes.data), arr.nbytes)
        self.sync()

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags.c_contiguous
        self._call("wm_memcpy_d2h", _vp(arr.ctypes.data), _vp(dptr), arr.nbytes)

    def copy_mapped(self, dst: int, src: int, nbytes: int, n_workgroups: int = 0):
        """async copy kernel between device memory and mapped pinned host memory (either direction) on this stream"""
        self._call("wm_copy_mapped_dev", _vp(dst), _vp(src), nbytes, n_workgroups)

    def memset(self, dptr: int, value: int, nbytes: int):
        self._call("wm_memset", _vp(dptr), value, nbytes)

    def event_record(self, slot: int):
        self._call("wm_event_record", slot)

    def event_elapsed_ms(self, a: int, b: int) -> float:
        ms = _f(0)
        self._call("wm_event_elapsed_ms", a, b, C.byref(ms))
        return ms.value

    # ---- device-pointer entry points (ints are device addresses) ---------
    def embed_tiles_u8_dev(self, host, sigma_w, stego, sigma_c, yw, n_planes, H, W, row_stride,
                           plane_stride, sigma_w_plane_stride, alpha, K):
        self._call("wm_embed_tiles_u8_dev", _vp(host), _vp(sigma_w), _vp(stego), _vp(sigma_c),
                   _vp(yw) if yw else None, n_planes, H, W, row_stride, plane_stride,
                   sigma_w_plane_stride, alpha, K)

    def sigma_tiles_u8_dev(self, planes, sigma, n_planes, H, W, row_stride, plane_stride):
        self._call("wm_sigma_tiles_u8_dev", _vp(planes), _vp(sigma), n_planes, H, W, row_stride, plane_stride)

    def svd_tiles_f32_dev(self, planes, U, S, Vt, n_planes, H, W, row_stride, plane_stride):
        self._call("wm_svd_tiles_f32_dev", _vp(planes), _vp(U), _vp(S), _vp(Vt), n_planes, H, W,
                   row_stride, plane_stride)

    def extract_tiles_u8_dev(self, stego, sigma_c, Uw, Vwt, out, n_planes, H, W, row_stride,
                             plane_stride, uv_plane_stride, alpha, K):
        self._call("wm_extract_tiles_u8_dev", _vp(stego), _vp(sigma_c), _vp(Uw), _vp(Vwt), _vp(out),
                   n_planes, H, W, row_stride, plane_stride, uv_plane_stride, alpha, K)

    def extract_tiles_px_u8_dev(self, stego, sigma_c, Ux, Vxt, out, n_planes, H, W, row_stride,
                                plane_stride, uv_plane_stride, alpha, K):
        """extract_tiles_u8_dev with pixel-domain factors (tile_factors_to_pixel_dev)."""
        self._call("wm_extract_tiles_px_u8_dev", _vp(stego), _vp(sigma_c), _vp(Ux), _vp(Vxt), _vp(out),
                   n_planes, H, W, row_stride, plane_stride, uv_plane_stride, alpha, K)

    def tile_factors_to_pixel_dev(self, Uw, Vwt, Ux, Vxt, n_tiles):
        self._call("wm_tile_factors_to_pixel_dev", _vp(Uw), _vp(Vwt), _vp(Ux), _vp(Vxt), n_tiles)

    def reconstruct_tiles_dev(self, Uw, sw_hat, Vwt, out, n_planes, H, W):
        self._call("wm_reconstruct_tiles_dev", _vp(Uw), _vp(sw_hat), _vp(Vwt), _vp(out), n_planes, H, W)

    def detect_tiles_u8_dev(self, stego, sigma_c, sigma_w, scores, n_planes, H, W, row_stride,
                            plane_stride, sigma_w_plane_stride, alpha):
        self._call("wm_detect_tiles_u8_dev", _vp(stego), _vp(sigma_c), _vp(sigma_w), _vp(scores),
                   n_planes, H, W, row_stride, plane_stride, sigma_w_plane_stride, alpha)

    # full-frame mode, device-resident planes / factors (ints are device addresses)
    def ref_embed_planes_u8_dev(self, host, sigma_w, stego, sigma_c, yw, n_planes, H, W, row_stride, plane_stride,
                                sigma_w_plane_stride, alpha, K):
        self._call("wm_ref_embed_planes_u8_dev", _vp(host), _vp(sigma_w), _vp(stego), _vp(sigma_c),
                   _vp(yw) if yw else None, n_planes, H, W, row_stride, plane_stride, sigma_w_plane_stride, alpha, K)

    def ref_sigma_planes_u8_dev(self, planes, sigma, n_planes, H, W, row_stride, plane_stride):
        self._call("wm_ref_sigma_planes_u8_dev", _vp(planes), _vp(sigma), n_planes, H, W, row_stride, plane_stride)

    def ref_extract_planes_u8_dev(self, stego, sigma_c, Uw, Vwt, out, n_planes, H, W, row_stride, plane_stride, alpha, K):
        self._call("wm_ref_extract_planes_u8_dev", _vp(stego), _vp(sigma_c), _vp(Uw), _vp(Vwt), _vp(out), n_planes,
                   H, W, row_stride, plane_stride, alpha, K)

    def ref_detect_planes_u8_dev(self, stego, sigma_c, sigma_w, scores, n_planes, H, W, row_stride, plane_stride, alpha):
        self._call("wm_ref_detect_planes_u8_dev", _vp(stego), _vp(sigma_c), _vp(sigma_w), _vp(scores), n_planes,
                   H, W, row_stride, plane_stride, alpha)

    # ---- NumPy (host-buffer) entry points --------------------------------
    def embed_tiles(self, host: np.ndarray, sigma_w: np.ndarray, alpha: float, K: int = 8,
                    want_yw: bool = False):
        """host uint8 [H,W] or [N,H,W]; sigma_w float32 [nby,nbx,8] (shared) or
        [N,nby,nbx,8].  Returns (stego uint8, sigma_c float32 [N?,nby,nbx,8], yw or None)."""
        if host.dtype != np.uint8:
            raise ValueError("host planes must be uint8")
        n, H, W, rs, ps = _plane_layout(host)
        nby, nbx = H // TILE, W // TILE
        sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        per_plane = sw.ndim == 4
        if sw.shape[-3:] != (nby, nbx, 8) or (per_plane and sw.shape[0] != n):
            raise ValueError(f"sigma_w shape {sw.shape} does not match {n}x{nby}x{nbx}x8")
        stego = np.empty((n, H, W), np.uint8)
        sc = np.empty((n, nby, nbx, 8), np.float32)
        yw = np.empty((n, H, W), np.float32) if want_yw else None
        # stego is dense; host may be strided -> densify so both share one layout
        hostc = np.ascontiguousarray(host.reshape(n, H, W))
        self._call("wm_embed_tiles_u8", _vp(hostc.ctypes.data), _vp(sw.ctypes.data), _vp(stego.ctypes.data),
                   _vp(sc.ctypes.data), _vp(yw.ctypes.data) if want_yw else None, n, H, W, W, H * W,
                   nby * nbx * 8 if per_plane else 0, float(alpha), int(K))
        if host.ndim == 2:
            return stego[0], sc[0], (yw[0] if want_yw else None)
        return stego, sc, yw

    def sigma_tiles(self, planes: np.ndarray) -> np.ndarray:
        if planes.dtype != np.uint8:
            raise ValueError("planes must be uint8")
        n, H, W, rs, ps = _plane_layout(planes)
        s = np.empty((n, H // TILE, W // TILE, 8), np.float32)
        self._call("wm_sigma_tiles_u8", _vp(planes.ctypes.data), _vp(s.ctypes.data), n, H, W, rs, ps)
        return s[0] if planes.ndim == 2 else s

    def svd_tiles(self, planes: np.ndarray):
        """float32 planes -> (U [..,nby,nbx,8,8], S [..,nby,nbx,8], Vt [..,nby,nbx,8,8])."""
        if planes.dtype != np.float32:
            raise ValueError("planes must be float32")
        n, H, W, rs, ps = _plane_layout(planes)
        nby, nbx = H // TILE, W // TILE
        U = np.empty((n, nby, nbx, 8, 8), np.float32)
        S = np.empty((n, nby, nbx, 8), np.float32)
        Vt = np.empty((n, nby, nbx, 8, 8), np.float32)
        self._call("wm_svd_tiles_f32", _vp(planes.ctypes.data), _vp(U.ctypes.data), _vp(S.ctypes.data),
                   _vp(Vt.ctypes.data), n, H, W, rs, ps)
        if planes.ndim == 2:
            return U[0], S[0], Vt[0]
        return U, S, Vt

    def extract_tiles(self, stego: np.ndarray, sigma_c: np.ndarray, Uw: np.ndarray, Vwt: np.ndarray,
                      alpha: float, K: int = 8, sum_planes: bool = False) -> np.ndarray:
        """Scrambled-watermark estimate per plane, float32 [n, H, W]; with ``sum_planes`` the planes
        are added on the device and only their sum [H, W] comes back (video extract averages frames)."""
        if stego.dtype != np.uint8:
            raise ValueError("stego planes must be uint8")
        n, H, W, rs, ps = _plane_layout(stego)
        nby, nbx = H // TILE, W // TILE
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32)
        if sc.size != n * nby * nbx * 8:
            raise ValueError(f"sigma_c has {sc.size} values, the planes need {n}x{nby}x{nbx}x8")
        sc = sc.reshape(n, nby, nbx, 8)
        Uw = np.ascontiguousarray(Uw, dtype=np.float32)
        Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        per_plane = Uw.ndim == 5
        if Uw.ndim not in (4, 5) or Uw.shape[-4:] != (nby, nbx, 8, 8) or Vwt.shape != Uw.shape \
                or (per_plane and Uw.shape[0] != n):
            raise ValueError(f"Uw/Vwt shape {Uw.shape}/{Vwt.shape} does not match {n} plane(s) of {nby}x{nbx} tiles")
        if sum_planes:
            tot = np.empty((H, W), np.float32)
            self._call("wm_extract_tiles_sum_u8", _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(Uw.ctypes.data),
                       _vp(Vwt.ctypes.data), _vp(tot.ctypes.data), n, H, W, rs, ps,
                       nby * nbx if per_plane else 0, float(alpha), int(K))
            return tot
        out = np.empty((n, H, W), np.float32)
        self._call("wm_extract_tiles_u8", _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(Uw.ctypes.data),
                   _vp(Vwt.ctypes.data), _vp(out.ctypes.data), n, H, W, rs, ps,
                   nby * nbx if per_plane else 0, float(alpha), int(K))
        return out[0] if stego.ndim == 2 else out

    def reconstruct_tiles(self, Uw: np.ndarray, sw_hat: np.ndarray, Vwt: np.ndarray, H: int, W: int):
        Uw = np.ascontiguousarray(Uw, dtype=np.float32)
        Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        sh = np.ascontiguousarray(sw_hat, dtype=np.float32)
        single = Uw.ndim == 4
        n = 1 if single else Uw.shape[0]
        nby, nbx = H // TILE, W // TILE
        if Uw.shape[-4:] != (nby, nbx, 8, 8) or Vwt.shape != Uw.shape or sh.size != n * nby * nbx * 8:
            raise ValueError("Uw / Vwt / sw_hat do not match the plane size")
        out = np.empty((n, H, W), np.float32)
        self._call("wm_reconstruct_tiles", _vp(Uw.ctypes.data), _vp(sh.ctypes.data), _vp(Vwt.ctypes.data),
                   _vp(out.ctypes.data), n, H, W)
        return out[0] if single else out

    def detect_tiles(self, stego: np.ndarray, sigma_c: np.ndarray, sigma_w: np.ndarray,
                     alpha: float) -> np.ndarray:
        if stego.dtype != np.uint8:
            raise ValueError("stego planes must be uint8")
        n, H, W, rs, ps = _plane_layout(stego)
        nby, nbx = H // TILE, W // TILE
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32)
        if sc.size != n * nby * nbx * 8:
            raise ValueError(f"sigma_c has {sc.size} values, the planes need {n}x{nby}x{nbx}x8")
        sc = sc.reshape(n, nby, nbx, 8)
        sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        per_plane = sw.ndim == 4
        if sw.ndim not in (3, 4) or sw.shape[-3:] != (nby, nbx, 8) or (per_plane and sw.shape[0] != n):
            raise ValueError(f"sigma_w shape {sw.shape} does not match {n} plane(s) of {nby}x{nbx} tiles")
        scores = np.zeros(n, np.float64)
        self._call("wm_detect_tiles_u8", _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(sw.ctypes.data),
                   _vp(scores.ctypes.data), n, H, W, rs, ps, nby * nbx * 8 if per_plane else 0, float(alpha))
        return scores

    # ---- full-frame mode (tile=None), one plane per call -----------------------
    def ref_embed(self, host: np.ndarray, sigma_w: np.ndarray, alpha: float, K: int, want_yw: bool = False):
        if host.dtype != np.uint8 or host.ndim != 2:
            raise ValueError("host plane must be uint8 [H, W]")
        host = np.ascontiguousarray(host)
        H, W = host.shape
        L = min(H, W)
        sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        if sw.shape != (L,):
            raise ValueError(f"sigma_w must have shape ({L},)")
        stego = np.empty_like(host); sc = np.empty(L, np.float32)
        yw = np.empty((H, W), np.float32) if want_yw else None
        self._call("wm_ref_embed_u8", _vp(host.ctypes.data), _vp(sw.ctypes.data), _vp(stego.ctypes.data),
                   _vp(sc.ctypes.data), _vp(yw.ctypes.data) if want_yw else None, H, W, W, float(alpha), int(K))
        return stego, sc, yw

    def ref_embed_planes(self, hosts: np.ndarray, sigma_w: np.ndarray, alpha: float, K: int, want_yw: bool = False):
        """hosts uint8 [N, H, W]; sigma_w [L] (shared) or [N, L].  All planes share every launch."""
        if hosts.dtype != np.uint8 or hosts.ndim != 3:
            raise ValueError("hosts must be uint8 [N, H, W]")
        hosts = np.ascontiguousarray(hosts)
        n, H, W = hosts.shape
        L = min(H, W)
        sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        if sw.shape not in ((L,), (n, L)):
            raise ValueError(f"sigma_w must have shape ({L},) or ({n}, {L})")
        stego = np.empty_like(hosts); sc = np.empty((n, L), np.float32)
        yw = np.empty((n, H, W), np.float32) if want_yw else None
        self._call("wm_ref_embed_planes_u8", _vp(hosts.ctypes.data), _vp(sw.ctypes.data), _vp(stego.ctypes.data),
                   _vp(sc.ctypes.data), _vp(yw.ctypes.data) if want_yw else None, n, H, W, W, H * W,
                   L if sw.ndim == 2 else 0, float(alpha), int(K))
        return stego, sc, yw

    def ref_embed_planes_when(self, hosts: np.ndarray, produce_sigma_w, per_plane: bool, alpha: float, K: int,
                              want_yw: bool = False):
        """``ref_embed_planes`` for a sigma_w that is still being computed: ``produce_sigma_w()`` -> (sigma_w [L] or [N, L],
        anything) runs on a worker thread - the watermark's own decomposition, on ANOTHER Context - while this context
        decomposes the host planes (single:172-173 are independent statements; one full-frame SVD leaves most of the chip
        idle).  Returns (stego, sigma_c, yw, the callable's second value); the callable's exception is re-raised here."""
        import threading
        if hosts.dtype != np.uint8 or hosts.ndim != 3:
            raise ValueError("hosts must be uint8 [N, H, W]")
        hosts = np.ascontiguousarray(hosts)
        n, H, W = hosts.shape
        L = min(H, W)
        sw = np.zeros((n, L) if per_plane else (L,), np.float32)
        flag = _i(0)
        box = {}

        def run():
            try:
                s, extra = produce_sigma_w()
                s = np.asarray(s, dtype=np.float32)
                if s.shape != sw.shape:
                    raise ValueError(f"sigma_w must have shape {sw.shape}")
                sw[...] = s
                box["extra"] = extra
                flag.value = 1                                   # read with acquire order by the library
            except BaseException as e:                           # the embed call returns WM_ERR_BADARG; re-raised below
                box["exc"] = e
                flag.value = -1
        t = threading.Thread(target=run, daemon=True)
        t.start()
        stego = np.empty_like(hosts); sc = np.empty((n, L), np.float32)
        yw = np.empty((n, H, W), np.float32) if want_yw else None
        try:
            self._call("wm_ref_embed_planes_u8_when", _vp(hosts.ctypes.data), _vp(sw.ctypes.data), C.byref(flag),
                       _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(yw.ctypes.data) if want_yw else None, n, H, W, W, H * W,
                       L if per_plane else 0, float(alpha), int(K))
        except Exception:
            t.join()
            if "exc" in box:
                raise box["exc"]
            raise
        t.join()
        return stego, sc, yw, box["extra"]

    def ref_last_sweeps(self) -> int:
        n = _i(0)
        self._call("wm_ref_last_sweeps", C.byref(n))
        return n.value

    def ref_last_flops(self):
        """(matrix-core flops the last full-frame SVD's Gram / rotation products issued, ran the two-level scheme?)"""
        f = C.c_double(0.0); h = _i(0)
        self._call("wm_ref_last_flops", C.byref(f), C.byref(h))
        return f.value, bool(h.value)

    def ref_sigma_planes(self, planes: np.ndarray) -> np.ndarray:
        if planes.dtype != np.uint8 or planes.ndim != 3:
            raise ValueError("planes must be uint8 [N, H, W]")
        planes = np.ascontiguousarray(planes)
        n, H, W = planes.shape
        s = np.empty((n, min(H, W)), np.float32)
        self._call("wm_ref_sigma_planes_u8", _vp(planes.ctypes.data), _vp(s.ctypes.data), n, H, W, W, H * W)
        return s

    def ref_sigma(self, plane: np.ndarray) -> np.ndarray:
        if plane.dtype != np.uint8 or plane.ndim != 2:
            raise ValueError("plane must be uint8 [H, W]")
        plane = np.ascontiguousarray(plane)
        H, W = plane.shape
        s = np.empty(min(H, W), np.float32)
        self._call("wm_ref_sigma_u8", _vp(plane.ctypes.data), _vp(s.ctypes.data), H, W, W)
        return s

    def ref_svd(self, plane: np.ndarray, apply_dct: bool = True):
        if plane.dtype != np.float32 or plane.ndim != 2:
            raise ValueError("plane must be float32 [H, W]")
        plane = np.ascontiguousarray(plane)
        H, W = plane.shape
        L = min(H, W)
        U = np.empty((H, L), np.float32); S = np.empty(L, np.float32); Vt = np.empty((L, W), np.float32)
        self._call("wm_ref_svd_f32", _vp(plane.ctypes.data), _vp(U.ctypes.data), _vp(S.ctypes.data),
                   _vp(Vt.ctypes.data), H, W, W, 1 if apply_dct else 0)
        return U, S, Vt

    def ref_svd_planes(self, planes: np.ndarray, apply_dct: bool = True):
        """planes float32 [n, H, W] -> U [n, H, L], S [n, L], Vt [n, L, W]: the watermark-side SVDs of a colour watermark's
        three planes (single:128-134) as one batch."""
        if planes.dtype != np.float32 or planes.ndim != 3:
            raise ValueError("planes must be float32 [n, H, W]")
        planes = np.ascontiguousarray(planes)
        n, H, W = planes.shape
        L = min(H, W)
        U = np.empty((n, H, L), np.float32); S = np.empty((n, L), np.float32); Vt = np.empty((n, L, W), np.float32)
        self._call("wm_ref_svd_planes_f32", _vp(planes.ctypes.data), _vp(U.ctypes.data), _vp(S.ctypes.data),
                   _vp(Vt.ctypes.data), n, H, W, W, H * W, 1 if apply_dct else 0)
        return U, S, Vt

    def ref_extract(self, stego: np.ndarray, sigma_c, Uw, Vwt, alpha: float, K: int) -> np.ndarray:
        if stego.dtype != np.uint8 or stego.ndim != 2:
            raise ValueError("stego plane must be uint8 [H, W]")
        stego = np.ascontiguousarray(stego)
        H, W = stego.shape
        L = min(H, W)
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32)
        Uw = np.ascontiguousarray(Uw, dtype=np.float32); Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        if sc.shape != (L,) or Uw.shape != (H, L) or Vwt.shape != (L, W):
            raise ValueError("meta arrays do not match the plane size")
        out = np.empty((H, W), np.float32)
        self._call("wm_ref_extract_u8", _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(Uw.ctypes.data),
                   _vp(Vwt.ctypes.data), _vp(out.ctypes.data), H, W, W, float(alpha), int(K))
        return out

    def ref_reconstruct(self, Uw, sw_hat, Vwt, H: int, W: int) -> np.ndarray:
        """single:214-218 with the estimates given: ``Uw[:L,:L] @ diag(sw_hat) @ Vwt[:L,:L]`` (L = len(sw_hat)) in the
        top-left corner of a zero H x W plane, then idct2.  Uw [H, min(H,W)], Vwt [min(H,W), W] as the meta holds them."""
        Lm = min(H, W)
        Uw = np.ascontiguousarray(Uw, dtype=np.float32); Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        sh = np.ascontiguousarray(sw_hat, dtype=np.float32).reshape(-1)
        if Uw.shape != (H, Lm) or Vwt.shape != (Lm, W):
            raise ValueError(f"Uw {Uw.shape} / Vwt {Vwt.shape} are not the factors of a {H}x{W} plane")
        if sh.size > Lm:
            raise ValueError(f"{sh.size} estimates for a plane with {Lm} singular values")
        out = np.empty((H, W), np.float32)
        self._call("wm_ref_reconstruct_f32", _vp(Uw.ctypes.data), _vp(sh.ctypes.data) if sh.size else _vp(out.ctypes.data),
                   _vp(Vwt.ctypes.data), _vp(out.ctypes.data), H, W, int(sh.size))
        return out

    def ref_extract_planes(self, stegos: np.ndarray, sigma_c, Uw, Vwt, alpha: float, K: int) -> np.ndarray:
        """stegos uint8 [n, H, W] sharing one watermark decomposition; sigma_c [n, L]."""
        if stegos.dtype != np.uint8 or stegos.ndim != 3:
            raise ValueError("stego planes must be uint8 [n, H, W]")
        stegos = np.ascontiguousarray(stegos)
        n, H, W = stegos.shape
        L = min(H, W)
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32)
        Uw = np.ascontiguousarray(Uw, dtype=np.float32); Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        if sc.shape != (n, L) or Uw.shape != (H, L) or Vwt.shape != (L, W):
            raise ValueError("meta arrays do not match the plane size")
        out = np.empty((n, H, W), np.float32)
        self._call("wm_ref_extract_planes_u8", _vp(stegos.ctypes.data), _vp(sc.ctypes.data), _vp(Uw.ctypes.data),
                   _vp(Vwt.ctypes.data), _vp(out.ctypes.data), n, H, W, W, H * W, float(alpha), int(K))
        return out

    def ref_detect(self, stego: np.ndarray, sigma_c, sigma_w, alpha: float) -> float:
        if stego.dtype != np.uint8 or stego.ndim != 2:
            raise ValueError("stego plane must be uint8 [H, W]")
        stego = np.ascontiguousarray(stego)
        H, W = stego.shape
        L = min(H, W)
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32); sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        # the C side reads min(H, W) floats from each buffer; the reference would truncate to the
        # shortest of Sc / S_cw / Sw (single:299,311-313) - a meta that does not belong to this
        # stego is refused here instead of being read past its end
        if sc.shape != (L,) or sw.shape != (L,):
            raise ValueError(f"sigma_c {sc.shape} / sigma_w {sw.shape} do not match the plane's {L} singular values")
        score = C.c_double(0.0)
        self._call("wm_ref_detect_u8", _vp(stego.ctypes.data), _vp(sc.ctypes.data), _vp(sw.ctypes.data),
                   C.byref(score), H, W, W, float(alpha))
        return score.value

    def ref_detect_planes(self, stegos: np.ndarray, sigma_c, sigma_w, alpha: float) -> np.ndarray:
        """stegos uint8 [n, H, W] carrying one watermark; sigma_c [n, L], sigma_w [L] -> scores float64 [n]."""
        if stegos.dtype != np.uint8 or stegos.ndim != 3:
            raise ValueError("stego planes must be uint8 [n, H, W]")
        stegos = np.ascontiguousarray(stegos)
        n, H, W = stegos.shape
        L = min(H, W)
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32); sw = np.ascontiguousarray(sigma_w, dtype=np.float32)
        if sc.shape != (n, L) or sw.shape != (L,):
            raise ValueError("meta arrays do not match the plane size")
        scores = np.empty(n, np.float64)
        self._call("wm_ref_detect_planes_u8", _vp(stegos.ctypes.data), _vp(sc.ctypes.data), _vp(sw.ctypes.data),
                   _vp(scores.ctypes.data), n, H, W, W, H * W, float(alpha))
        return scores

    # ---- keyed scramble / unscramble on the device (single:66-80) ------------------
    # The permutation is NumPy's own PCG64 shuffle (hostglue.permutation_index, bit-exact by construction);
    # its int32 copy is uploaded once per index array and kept (two entries, like the host-side cache).
    ROUTE_MAX_N = 2048 << 15        # wm_route: at most 2048 blocks of 32768 elements (67 M pixels, an 8K x 8K plane)

    def _drop_index(self, ent):
        try:
            if ent.get("route"):
                self.lib.wm_route_destroy(self._h, _vp(ent["route"]))
            self.free(ent["d"])
        except Exception:
            pass

    def _index_entry(self, idx: np.ndarray) -> dict:
        """Device copy (int32) of a permutation index.  Cached on the index's identity: hostglue.permutation_index
        tags its result with (H, W, sha256(key)); any other array is keyed by a digest of its contents - never by its
        address, which NumPy reuses for the next index of the same size."""
        cache = self.__dict__.setdefault("_idx_cache", [])
        tag = getattr(idx, "tag", None)
        trusted = tag is not None                  # hostglue's own shuffle of arange: a bijection by construction
        if tag is None:
            tag = ("digest", idx.size, hashlib.blake2b(np.ascontiguousarray(idx).view(np.uint8), digest_size=16).digest())
        for ent in cache:
            if ent["tag"] == tag:
                return ent
        if idx.size > 0x7fffffff:
            raise ValueError("plane too large for an int32 index")
        if idx.size and (int(idx.min()) < 0 or int(idx.max()) >= idx.size):
            raise ValueError("index entries must lie in [0, n)")      # the device passes gather / scatter through it unchecked
        if not trusted and idx.size:
            seen = np.zeros(idx.size, np.bool_)
            seen[np.asarray(idx)] = True
            if not seen.all():
                raise ValueError("index is not a permutation (repeated entries): the inverse scatter would leave holes")
        i32 = np.ascontiguousarray(idx, dtype=np.int32)
        d = self.malloc(max(i32.nbytes, 4))
        self.h2d(d, i32)
        ent = {"tag": tag, "d": d, "n": int(idx.size), "route": None}
        cache.append(ent)
        while len(cache) > 2:
            old = cache.pop(0)
            self.sync()
            self._drop_index(old)
        return ent

    def index_dev(self, idx: np.ndarray) -> int:
        return self._index_entry(idx)["d"]

    def route_dev(self, idx: np.ndarray):
        """The wm_route of this permutation (built once per key, kept with the device index), or None when the
        plane is empty or has more elements than a route addresses."""
        ent = self._index_entry(idx)
        if ent["route"] is None and 0 < ent["n"] <= self.ROUTE_MAX_N:
            r = _vp()
            self._call("wm_route_create_dev", _vp(ent["d"]), ent["n"], C.byref(r))
            ent["route"] = r.value
        return ent["route"]

    def permute_planes(self, planes: np.ndarray, idx: np.ndarray) -> np.ndarray:
        """``flat[idx]`` of every plane (uint8 or float32 [H, W] / [n, H, W]) -> float32, like hostglue.permute."""
        single = planes.ndim == 2
        p = np.ascontiguousarray(planes[None] if single else planes)
        n_pl, H, W = p.shape
        n = H * W
        if idx.size != n:
            raise ValueError("index length does not match the plane")
        d_idx = self.index_dev(idx)
        d_src = self.malloc(p.nbytes); d_dst = self.malloc(n_pl * n * 4)
        try:
            self.h2d(d_src, p)
            route = self.route_dev(idx) if p.dtype == np.uint8 else None
            if route is not None:          # the coalesced two-pass form (csrc/wm_route.hip), same values
                self._call("wm_permute_u8_f32_routed_dev", _vp(d_src), _vp(route), _vp(d_dst), n, n_pl)
            elif p.dtype == np.uint8:
                self._call("wm_permute_u8_f32_dev", _vp(d_src), _vp(d_idx), _vp(d_dst), n, n_pl)
            elif p.dtype == np.float32:
                self._call("wm_permute_f32_dev", _vp(d_src), _vp(d_idx), _vp(d_dst), n, n_pl)
            else:
                raise ValueError("planes must be uint8 or float32")
            out = np.empty((n_pl, H, W), np.float32)
            self.d2h(out, d_dst)
        finally:
            self.free(d_src); self.free(d_dst)
        return out[0] if single else out

    def unpermute_normalize_u8(self, planes: np.ndarray, idx: np.ndarray, normalize: bool = True) -> np.ndarray:
        """hostglue.unpermute + the reference's min-max normalise / clip / uint8 (single:220-222) per plane, on the
        device: float32 [H, W] / [n, H, W] in, uint8 out."""
        single = planes.ndim == 2
        p = np.ascontiguousarray(planes[None] if single else planes, dtype=np.float32)
        n_pl, H, W = p.shape
        d_src = self.malloc(p.nbytes)
        try:
            self.h2d(d_src, p)
            out = self._unpermute_normalize_dev(d_src, n_pl, H, W, idx, normalize)
        finally:
            self.free(d_src)
        return out[0] if single else out

    def _unpermute_normalize_dev(self, d_src: int, n_pl: int, H: int, W: int, idx: np.ndarray, normalize: bool) -> np.ndarray:
        """d_src: float32 [n_pl][H*W] on the device -> uint8 [n_pl, H, W] on the host (single:74-80, 221-222; every plane
        is normalised on its own, single:269-274).  Routed (coalesced two-pass) form whenever the plane fits a route."""
        n = H * W
        if idx.size != n:
            raise ValueError("index length does not match the plane")
        out = np.empty((n_pl, H, W), np.uint8)
        route = self.route_dev(idx)
        if route is not None:
            d_u8 = self.malloc(max(n_pl * n, 16))
            try:
                self._call("wm_unpermute_normalize_u8_dev", _vp(d_src), _vp(route), _vp(d_u8), n, n_pl, 1 if normalize else 0)
                self.d2h(out, d_u8)
            finally:
                self.free(d_u8)
            return out
        d_idx = self.index_dev(idx)
        n_pad = (n + 3) & ~3                      # the normalise kernel reads float4: every plane starts 16-byte aligned
        d_tmp = self.malloc(n_pl * n_pad * 4); d_u8 = self.malloc(n_pl * n_pad)
        try:
            for z in range(n_pl):
                self._call("wm_unpermute_f32_dev", _vp(d_src + z * n * 4), _vp(d_idx), _vp(d_tmp + z * n_pad * 4), n, 1)
                self._call("wm_normalize_u8_dev", _vp(d_tmp + z * n_pad * 4), n, 1 if normalize else 0,
                           _vp(d_u8 + z * n_pad))
                self.d2h(out[z], d_u8 + z * n_pad)
        finally:
            self.free(d_tmp); self.free(d_u8)
        return out

    def extract_tiles_unscrambled_u8(self, stego: np.ndarray, sigma_c: np.ndarray, Uw: np.ndarray, Vwt: np.ndarray,
                                     alpha: float, K: int, idx: np.ndarray, normalize: bool = True) -> np.ndarray:
        """extract_tiles + unpermute + normalise without the float plane ever leaving the device
        (single:204-222 / 232-274): uint8 stego planes in, uint8 watermark planes out."""
        if stego.dtype != np.uint8:
            raise ValueError("stego planes must be uint8")
        single = stego.ndim == 2
        st = np.ascontiguousarray(stego[None] if single else stego)
        n, H, W = st.shape
        nby, nbx = H // TILE, W // TILE
        sc = np.ascontiguousarray(sigma_c, dtype=np.float32)
        Uw = np.ascontiguousarray(Uw, dtype=np.float32); Vwt = np.ascontiguousarray(Vwt, dtype=np.float32)
        per_plane = Uw.ndim == 5
        if sc.size != n * nby * nbx * 8 or Uw.shape[-4:] != (nby, nbx, 8, 8) or Vwt.shape != Uw.shape \
                or (per_plane and Uw.shape[0] != n):
            raise ValueError("meta arrays do not match the planes")
        route = self.route_dev(idx) if idx.size == H * W else None
        bufs = [self.malloc(max(x, 16)) for x in (st.nbytes, sc.nbytes, Uw.nbytes, Vwt.nbytes,
                                                   n * H * W if route is not None else n * H * W * 4)]
        d_st, d_sc, d_u, d_v, d_w = bufs
        try:
            self.h2d(d_st, st); self.h2d(d_sc, sc); self.h2d(d_u, Uw); self.h2d(d_v, Vwt)
            if route is not None:          # one call: extract kernel (with its min / max) -> routed unscramble -> uint8
                self._call("wm_extract_unscrambled_u8_dev", _vp(d_st), _vp(d_sc), _vp(d_u), _vp(d_v), _vp(route), _vp(d_w),
                           n, H, W, W, H * W, nby * nbx if per_plane else 0, float(alpha), int(K), 0, 1 if normalize else 0)
                out = np.empty((n, H, W), np.uint8)
                self.d2h(out, d_w)
            else:
                self.extract_tiles_u8_dev(d_st, d_sc, d_u, d_v, d_w, n, H, W, W, H * W, nby * nbx if per_plane else 0,
                                          float(alpha), int(K))
                out = self._unpermute_normalize_dev(d_w, n, H, W, idx, normalize)
            # the *_dev entry points only set the sticky status bit; without this a Jacobi that hit its sweep
            # bound would hand back a watermark silently and surface in some later, unrelated call (DESIGN 5:
            # non-convergence -> WM_ERR_NOCONV -> numpy.linalg.LinAlgError, like the host-pointer wrapper)
            self.check_status()
        finally:
            for b in bufs:
                self.free(b)
        return out[0] if single else out

    # ---- pixel-side kernels (colour, PSNR, SSIM, normalise) ----------------------
    _COLOR_OPS = {"bgr2ycrcb": 0, "ycrcb2bgr": 1, "bgr2gray": 2, "bgr2y": 3, "replace_y": 4}

    def color(self, op: str, img: np.ndarray, plane: Optional[np.ndarray] = None) -> np.ndarray:
        """img: uint8 [H, W, 3]; returns [H, W, 3] (ops 0, 1, 4) or the [H, W] plane (ops 2, 3)."""
        code = self._COLOR_OPS[op]
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("img must be uint8 [H, W, 3]")
        img = np.ascontiguousarray(img)
        H, W = img.shape[:2]
        if code in (2, 3):
            out = np.empty((H, W), np.uint8)
            self._call("wm_color_u8", code, _vp(img.ctypes.data), None, None, _vp(out.ctypes.data), H * W)
            return out
        out = np.empty_like(img)
        pin = None
        if code == 4:
            plane = np.ascontiguousarray(plane, dtype=np.uint8)
            if plane.shape != (H, W):
                raise ValueError("plane must be [H, W]")
            pin = _vp(plane.ctypes.data)
        self._call("wm_color_u8", code, _vp(img.ctypes.data), pin, _vp(out.ctypes.data), None, H * W)
        return out

    def psnr(self, a: np.ndarray, b: np.ndarray) -> float:
        a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        v = C.c_double(0.0)
        self._call("wm_psnr_u8", _vp(a.ctypes.data), _vp(b.ctypes.data), a.size, C.byref(v))
        return v.value

    def ssim(self, img1: np.ndarray, img2: np.ndarray) -> float:
        """planes [H, W], each uint8 or float32"""
        def prep(x):
            if x.dtype == np.uint8:
                return np.ascontiguousarray(x), 0
            return np.ascontiguousarray(x, dtype=np.float32), 1
        x, k1 = prep(img1); y, k2 = prep(img2)
        if x.shape != y.shape or x.ndim != 2:
            raise ValueError("planes must be [H, W] of equal shape")
        v = C.c_double(0.0)
        self._call("wm_ssim", _vp(x.ctypes.data), _vp(y.ctypes.data), x.shape[0], x.shape[1], k1 | (k2 << 1), C.byref(v))
        return v.value

    def normalize_u8(self, x: np.ndarray, normalize: bool = True) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.shape, np.uint8)
        self._call("wm_normalize_u8", _vp(x.ctypes.data), x.size, 1 if normalize else 0, _vp(out.ctypes.data))
        return out
