"""Drop-in for the reference's ``embed / extract / detect`` surface
(app_dct_svd_single.py:112-318 - the authoritative "SECURE CORE"; imported by
app_dct_svd_pyside6.py:8 as ``from dct_svd_core_secure import embed, extract,
detect``), with the numeric hot path on MI355X HIP kernels.

Same positional/keyword arguments, return values, side effects (``_stego.png``
/ ``_wm.png`` renaming, ``.npz`` meta with the same keys) and exception types
as the reference.  Keyword-only extras, all with reference-preserving
defaults where the reference has a behaviour:

  tile     8 (default): the north_star's block formulation - the reference's
           per-matrix arithmetic applied to every 8x8 tile.
           Its meta carries ``tile=8`` and is not readable by the reference's
           own full-frame extract.  ``tile=None``: the reference's own
           full-frame semantics (one DCT + one dense SVD per plane) on the
           GPU; stego + meta written this way are what the reference's
           extract/detect expect (same keys, shapes, HMAC coverage).  Neither
           mode has a CPU fallback.
  k_floor  the literal 8 of ``K = max(8, int(kfrac*L))`` (single:174); at
           tile=8 the formula is 8 for every kfrac, so a mid-band sweep sets
           k_floor < 8.
  nonce    inject the 8-byte nonce (reference: ``os.urandom(8)``, single:119).
  device   HIP device index.

``embed_watermark`` / ``extract_watermark`` are aliases (the names
BASELINE.json's north_star uses).
"""
from __future__ import annotations

import os
import threading
from typing import Optional

import numpy as np

from . import hostapi
from . import hostglue as hg

K_FRAC_DEFAULT = hg.K_FRAC_DEFAULT
TILE = 8

_tls = threading.local()


def _ctx(device: int = 0, companion: bool = False) -> hostapi.Context:
    """One cached context per (calling thread, device): a context owns its stream and grow-only
    scratch buffers, so two threads inside embed()/extract() at once must not share one (the
    reference's functions are re-entrant).  Released when the thread ends.
    companion: this thread's second context, handed to the worker that decomposes the watermark while the
    first one decomposes the host planes (full-frame embed)."""
    cache = getattr(_tls, "contexts", None)
    if cache is None:
        cache = _tls.contexts = {}
    key = (device, bool(companion))
    c = cache.get(key)
    if c is None:
        c = cache[key] = hostapi.Context(device)
    return c


def _k_of(L: int, kfrac: float, k_floor: int) -> int:
    return min(L, max(int(k_floor), int(kfrac * L)))     # single:174 (capped at L = 8)


def _check_password(password, what: str) -> None:
    """The authoritative signatures (single:112-114,192) take the password as a string.  The legacy module of the
    same name had ``extract(stego, meta, out, normalize=True)`` and ``embed(..., payload_type, text_data)`` without one
    (core:85-92,203): a positional ``True`` from such a call site would otherwise be hashed as a password."""
    if password is not None and not isinstance(password, str):
        raise TypeError(f"password must be a str, got {type(password).__name__}: {what}(...) follows "
                        "app_dct_svd_single.py's signature (password before normalize / kfrac), not the legacy "
                        "dct_svd_core_secure.py one - pass password= and normalize= by keyword")


def _check_tile(tile):
    if tile is not None and int(tile) != TILE:
        raise ValueError("tile must be 8 or None")


# ---------------------------------------------------------------------------
# array level (decoded images in, arrays out) - what the file level wraps
# ---------------------------------------------------------------------------
class _Later:
    """A host-only computation (the HMAC over the meta's factors: 28 ms per 66 MB at 4K, the largest single item of a
    tile-mode call) on a worker thread while this thread drives the device; hashlib and ctypes both release the GIL."""

    def __init__(self, fn):
        import threading
        self._out = self._exc = None

        def run():
            try:
                self._out = fn()
            except BaseException as e:            # re-raised in result()
                self._exc = e
        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def result(self):
        self._t.join()
        if self._exc is not None:
            raise self._exc
        return self._out


def embed_arrays(cover: np.ndarray, wm: np.ndarray, password: str, nonce: bytes,
                 alpha: float = 0.1, color: bool = False, kfrac: float = K_FRAC_DEFAULT,
                 tile: Optional[int] = TILE, k_floor: int = 8, device: int = 0) -> dict:
    """cover, wm: BGR uint8.  Returns dict(stego BGR uint8, meta dict, psnr, ssim)."""
    _check_password(password, "embed")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để nhúng.")              # single:115-116
    _check_tile(tile)
    ctx = _ctx(device)
    H, W = cover.shape[:2]
    wm = hg.resize_area_cached(wm, W, H)                                   # single:118
    key = hg.derive_key(password, nonce)                                   # single:119
    idx = hg.permutation_index(H, W, key)
    if tile is None:
        return _embed_arrays_fullframe(ctx, cover, wm, key, idx, nonce, alpha, color, kfrac, k_floor)
    K = _k_of(TILE, kfrac, k_floor)
    common = dict(payload_type="image", shape=np.array((H, W)), alpha=float(alpha),
                  kfrac=float(kfrac), nonce=np.frombuffer(nonce, dtype=np.uint8),
                  tile=np.int32(TILE), k_floor=np.int32(k_floor))
    if color:
        hosts = np.ascontiguousarray(np.moveaxis(cover, -1, 0))            # b, g, r planes  single:122
        wms = ctx.permute_planes(np.ascontiguousarray(np.moveaxis(wm, -1, 0)), idx)              # single:123-126 (index pass on the device)
        U, S, Vt = ctx.svd_tiles(wms)                                      # single:131-134
        stego_p, Sc, _ = ctx.embed_tiles(hosts, S, alpha, K)               # single:127-147
        stego = np.ascontiguousarray(np.moveaxis(stego_p, 0, -1))
        meta = dict(mode="color", **common)
        for ch, n in enumerate("bgr"):
            meta["S" + n] = Sc[ch]; meta["UW" + n] = U[ch]
            meta["VW" + n + "t"] = Vt[ch]; meta["SW" + n] = S[ch]
        dg = _Later(lambda: hg.hmac_digest(key, [meta["Sb"], meta["Sg"], meta["Sr"],
                                                 meta["UWb"], meta["UWg"], meta["UWr"],
                                                 meta["VWbt"], meta["VWgt"], meta["VWrt"]]))   # single:152-156, under the metrics below
        ps = ctx.psnr(cover, stego)
        ss = ctx.ssim(ctx.color("bgr2gray", cover), ctx.color("bgr2gray", stego))               # single:167
        meta["digest"] = np.frombuffer(dg.result(), dtype=np.uint8)
        return dict(stego=stego, meta=meta, psnr=ps, ssim=ss)
    Y = ctx.color("bgr2y", cover)                                          # single:169  (_to_Y)
    wy_s = ctx.permute_planes(ctx.color("bgr2gray", wm), idx)              # single:170-171 (index pass on the device)
    Uw, Sw, Vwt = ctx.svd_tiles(wy_s)                                      # single:173
    stegoY, Sc, Yw = ctx.embed_tiles(Y, Sw, alpha, K, want_yw=True)        # single:172-177
    dg = _Later(lambda: hg.hmac_digest(key, [Sc, Uw, Vwt]))               # single:182, under the colour conversion and the metrics
    stego = ctx.color("replace_y", cover, stegoY)                          # single:26-30 (_from_Y)
    ps = ctx.psnr(cover, stego)
    ss = ctx.ssim(ctx.color("bgr2gray", cover), Yw)                        # single:190
    meta = dict(mode="gray", Sc=Sc, Uw=Uw, Vwt=Vwt, Sw=Sw, **common,
                digest=np.frombuffer(dg.result(), dtype=np.uint8))         # single:183-189
    return dict(stego=stego, meta=meta, psnr=ps, ssim=ss)


def _embed_arrays_fullframe(ctx, cover, wm, key, idx, nonce, alpha, color, kfrac, k_floor) -> dict:
    """tile=None: exactly the reference's meta (keys of single:157-166,183-189, no extras
    unless k_floor differs from the literal 8)."""
    H, W = cover.shape[:2]
    L = min(H, W)
    K = _k_of(L, kfrac, k_floor)
    common = dict(payload_type="image", shape=np.array((H, W)), alpha=float(alpha),
                  kfrac=float(kfrac), nonce=np.frombuffer(nonce, dtype=np.uint8))
    if k_floor != 8:
        common["k_floor"] = np.int32(k_floor)
    if color:
        meta = dict(mode="color", **common)
        Sws = []
        # single:123-134: the three scrambled watermark planes, their DCTs and SVDs as ONE batch - on the companion context
        # and a worker thread, under the host planes' own decomposition (the two do not depend on each other until
        # single:139's S + alpha * Sw)
        ctx_w = _ctx(ctx.device, companion=True)

        def watermark_side():
            w_s = ctx_w.permute_planes(np.ascontiguousarray(np.moveaxis(wm, -1, 0)), idx)      # one shared permutation, single:124-126
            Us, Ss, Vts = ctx_w.ref_svd_planes(w_s, apply_dct=True)
            return Ss, (Us, Ss, Vts)
        hosts = np.ascontiguousarray(np.moveaxis(cover, -1, 0))            # b, g, r planes, one batched call
        st, Sc, _, (Us, Ss, Vts) = ctx.ref_embed_planes_when(hosts, watermark_side, True, alpha, K)   # single:127-147
        for ch, n in enumerate("bgr"):
            meta["UW" + n] = np.ascontiguousarray(Us[ch]); meta["VW" + n + "t"] = np.ascontiguousarray(Vts[ch]); meta["SW" + n] = np.ascontiguousarray(Ss[ch])
            Sws.append(meta["SW" + n])
        for ch, n in enumerate("bgr"):
            meta["S" + n] = Sc[ch]
        dg = _Later(lambda: hg.hmac_digest(key, [meta["Sb"], meta["Sg"], meta["Sr"], meta["UWb"], meta["UWg"], meta["UWr"],
                                                 meta["VWbt"], meta["VWgt"], meta["VWrt"]]))   # single:152-156 (39 MB at 1080p: 16 ms), under the interleave and the metrics
        stego = np.ascontiguousarray(np.moveaxis(st, 0, -1))
        ps = ctx.psnr(cover, stego)
        ss = ctx.ssim(ctx.color("bgr2gray", cover), ctx.color("bgr2gray", stego))
        meta["digest"] = np.frombuffer(dg.result(), dtype=np.uint8)
        return dict(stego=stego, meta=meta, psnr=ps, ssim=ss)
    ctx_w = _ctx(ctx.device, companion=True)

    def watermark_side():                                                  # single:170-171, 173 on the companion context
        wy_s = ctx_w.permute_planes(ctx_w.color("bgr2gray", wm), idx)
        Uw, Sw, Vwt = ctx_w.ref_svd(wy_s, apply_dct=True)
        return Sw, (Uw, Sw, Vwt)
    Y = ctx.color("bgr2y", cover)
    st, Scs, Yws, (Uw, Sw, Vwt) = ctx.ref_embed_planes_when(Y[None], watermark_side, False, alpha, K, want_yw=True)   # single:172-177
    stegoY, Sc, Yw = st[0], Scs[0], Yws[0]
    dg = _Later(lambda: hg.hmac_digest(key, [Sc, Uw, Vwt]))               # single:182, under the colour conversion and the metrics
    stego = ctx.color("replace_y", cover, stegoY)
    ps = ctx.psnr(cover, stego)
    ss = ctx.ssim(ctx.color("bgr2gray", cover), Yw)
    meta = dict(mode="gray", Sc=Sc, Uw=Uw, Vwt=Vwt, Sw=Sw, **common,
                digest=np.frombuffer(dg.result(), dtype=np.uint8))
    return dict(stego=stego, meta=meta, psnr=ps, ssim=ss)


def _meta_tile(meta) -> Optional[int]:
    """Tile size a meta was written with: the explicit ``tile`` key, else inferred
    from the singular-value array (per-tile [nby, nbx, 8] vs full-frame [L])."""
    if "tile" in meta:
        t = int(meta["tile"])
        if t == 0:
            return None
        if t != TILE:
            raise ValueError("tile must be 8 or None")
        return TILE
    s = meta["Sc"] if "Sc" in meta else meta["Sb"]
    return TILE if np.asarray(s).ndim == 3 else None


def _check_stego_shape(stego: np.ndarray, meta):
    """Tile mode: a meta belongs to one stego size (per-tile factors; the mismatch is named before any
    device call).  Full-frame mode follows the reference, which goes on with the shortest of the
    lengths involved (single:210, 299) - see _extract_resized / _detect_resized."""
    H, W = map(int, meta["shape"])
    if tuple(stego.shape[:2]) != (H, W):
        raise ValueError(f"stego is {stego.shape[1]}x{stego.shape[0]} but the meta was written for {W}x{H}")


def _same_size(stego: np.ndarray, meta) -> bool:
    H, W = map(int, meta["shape"])
    return tuple(stego.shape[:2]) == (H, W)


def _extract_plane_resized(ctx, plane_u8, Sc, Uw, Vwt, alpha, kfrac, k_floor, H, W):
    """single:205-218 for a stego plane whose size is not the meta's (a resized or cropped stego): the
    reference does not look at the size - sigma of whatever plane it was given, L = the shortest of the four
    lengths, the [:L,:L] corner of the meta's factors, the META's H x W for the zero plane and the permutation."""
    S_cw = ctx.ref_sigma(plane_u8)                                         # single:205
    Sc = np.asarray(Sc, dtype=np.float32)
    L = min(len(Sc), len(S_cw), Uw.shape[0], Vwt.shape[0])                 # single:210
    K = _k_of(L, kfrac, k_floor)                                           # single:211
    sw_hat = ((S_cw[:L] - Sc[:L]) / np.float32(max(alpha, 1e-8))).astype(np.float32)   # single:212
    sw_hat[K:] = 0                                                         # single:213
    return ctx.ref_reconstruct(Uw, sw_hat, Vwt, H, W)                      # single:214-218


def _nc(a, b) -> float:
    """single:284-289"""
    a = np.asarray(a, dtype=np.float32).reshape(-1); b = np.asarray(b, dtype=np.float32).reshape(-1)
    if a.size == 0 or b.size == 0:
        return 0.0                                                         # single:286
    a = a - a.mean(); b = b - b.mean()
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-8))


def _detect_plane_resized(ctx, plane_u8, Sc, Sw, alpha) -> float:
    """single:297-301 when the three vectors differ in length: truncated to the shortest (single:299)."""
    S_cw = ctx.ref_sigma(plane_u8)
    Sc = np.asarray(Sc, dtype=np.float32).reshape(-1); Sw = np.asarray(Sw, dtype=np.float32).reshape(-1)
    L = min(len(Sc), len(S_cw), len(Sw))
    return _nc(Sw[:L], (S_cw[:L] - Sc[:L]) / np.float32(max(alpha, 1e-8)))


def extract_arrays(stego: np.ndarray, meta, password: str, normalize: bool = True,
                   device: int = 0) -> np.ndarray:
    """Watermark estimate (uint8 [H,W] gray / [H,W,3] colour) before the
    reference's cosmetic denoise/enhance step (single:223-227,275-277)."""
    _check_password(password, "extract")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để giải trích.")          # single:193-194
    mode = str(meta["mode"]); alpha = float(meta["alpha"])                 # single:196
    H, W = map(int, meta["shape"])
    nonce = bytes(bytearray(np.asarray(meta["nonce"]).astype(np.uint8).tolist()))
    digest = bytes(bytearray(np.asarray(meta["digest"]).astype(np.uint8).tolist()))
    key = hg.derive_key(password, nonce)                                   # single:200
    kfrac = float(meta["kfrac"]) if "kfrac" in meta else K_FRAC_DEFAULT    # single:211
    k_floor = int(meta["k_floor"]) if "k_floor" in meta else 8
    if mode == "gray":
        parts = [meta["Sc"], meta["Uw"], meta["Vwt"]]
    else:
        parts = [meta["S" + n] for n in "bgr"] + [meta["UW" + n] for n in "bgr"] \
            + [meta["VW" + n + "t"] for n in "bgr"]
    # single:206-209,244-247: the HMAC check runs on a worker thread UNDER the device work (it is 28 ms per 66 MB of factors,
    # more than everything else of a tile-mode extract); nothing is returned before it has passed, and a mismatch takes
    # precedence over whatever else went wrong meanwhile, as in the reference, where it comes first.  The overlap is only
    # taken for a key whose permutation is already cached (the extract that follows an embed, the frames of a clip): for any
    # other key the check is joined BEFORE the expensive key-dependent steps - the PCG64 shuffle of H*W indices, the route
    # build, the factor upload - so that a wrong password or a tampered meta costs one HMAC, evicts nothing from the
    # permutation / device-index caches and never reaches the native code (which would otherwise see unauthenticated
    # factors guarded by its shape checks alone).
    check = _Later(lambda: hg.digests_equal(hg.hmac_digest(key, parts), digest))
    if not hg.permutation_is_cached(H, W, key) and not check.result():
        raise ValueError("Sai mật khẩu hoặc meta không khớp.")             # single:208-209,246-247
    try:
        out = _extract_checked(stego, meta, mode, alpha, kfrac, k_floor, H, W, key, normalize, device)
    except BaseException:
        if not check.result():
            raise ValueError("Sai mật khẩu hoặc meta không khớp.") from None
        raise
    if not check.result():
        raise ValueError("Sai mật khẩu hoặc meta không khớp.")             # single:208-209,246-247
    return out


def _extract_checked(stego, meta, mode, alpha, kfrac, k_floor, H, W, key, normalize, device):
    tile = _meta_tile(meta)
    if tile is not None:
        _check_stego_shape(stego, meta)
    ctx = _ctx(device)
    idx = hg.permutation_index(H, W, key)                                  # single:219,265
    if tile is None:
        return _extract_arrays_fullframe(ctx, stego, meta, mode, alpha, kfrac, k_floor, H, W, idx, normalize)
    K = _k_of(TILE, kfrac, k_floor)
    if mode == "gray":
        Y = ctx.color("bgr2y", stego)                                      # single:204
        # single:205-222 in one device-resident chain: sigma -> rank-8 product -> unscramble -> normalise -> uint8
        return ctx.extract_tiles_unscrambled_u8(Y, meta["Sc"], meta["Uw"], meta["Vwt"], alpha, K, idx, normalize)
    planes = np.ascontiguousarray(np.moveaxis(stego, -1, 0))               # single:232
    Sc = np.stack([meta["S" + n] for n in "bgr"])
    U = np.stack([meta["UW" + n] for n in "bgr"])
    Vt = np.stack([meta["VW" + n + "t"] for n in "bgr"])
    ws = ctx.extract_tiles_unscrambled_u8(planes, Sc, U, Vt, alpha, K, idx, normalize)   # single:233-274
    return np.ascontiguousarray(np.moveaxis(ws, 0, -1))


def _extract_arrays_fullframe(ctx, stego, meta, mode, alpha, kfrac, k_floor, H, W, idx, normalize):
    def k_for(Sc, S_len_u, S_len_v):
        L = min(len(Sc), min(H, W), S_len_u, S_len_v)                      # single:210
        return _k_of(L, kfrac, k_floor)
    same = _same_size(stego, meta)            # a stego of another size: the reference's truncation rules, single:210
    if mode == "gray":
        Y = ctx.color("bgr2y", stego)
        Uw, Vwt = meta["Uw"], meta["Vwt"]
        if same:
            wy_s = ctx.ref_extract(Y, meta["Sc"], Uw, Vwt, alpha, k_for(meta["Sc"], Uw.shape[0], Vwt.shape[0]))
        else:
            wy_s = _extract_plane_resized(ctx, Y, meta["Sc"], Uw, Vwt, alpha, kfrac, k_floor, H, W)
        return ctx.unpermute_normalize_u8(wy_s, idx, normalize)
    outs = []
    # single:232-236: the three stego planes' singular values in ONE batched call (a single full-frame SVD is latency-bound on
    # a fraction of the chip: 3 planes cost 1.4 x one, not 3 x), then single:248-264 per channel with its own factors
    planes = np.ascontiguousarray(np.moveaxis(stego, -1, 0))
    S_cw = ctx.ref_sigma_planes(planes)
    for ch, n in enumerate("bgr"):
        U, Vt = meta["UW" + n], meta["VW" + n + "t"]
        Sc = np.asarray(meta["S" + n], dtype=np.float32)
        L = min(len(Sc), S_cw.shape[1], U.shape[0], Vt.shape[0])           # single:248
        K = _k_of(L, kfrac, k_floor)                                       # single:249
        sw_hat = ((S_cw[ch, :L] - Sc[:L]) / np.float32(max(alpha, 1e-8))).astype(np.float32)   # single:250-252
        sw_hat[K:] = 0
        w_s = ctx.ref_reconstruct(U, sw_hat, Vt, H, W)                     # single:257-264
        outs.append(ctx.unpermute_normalize_u8(w_s, idx, normalize))
    return np.stack(outs, axis=-1)


def detect_arrays(stego: np.ndarray, meta, thresh: float = 0.6, device: int = 0):
    mode = str(meta["mode"]); alpha = float(meta["alpha"])                 # single:293
    tile = _meta_tile(meta)
    if tile is not None:
        _check_stego_shape(stego, meta)
    ctx = _ctx(device)
    if tile is None:
        same = _same_size(stego, meta)        # another size: the vectors are cut to the shortest, single:299,311-313
        if mode == "gray":
            Y = ctx.color("bgr2y", stego)
            score = (ctx.ref_detect(Y, meta["Sc"], meta["Sw"], alpha) if same                 # single:297-301
                     else _detect_plane_resized(ctx, Y, meta["Sc"], meta["Sw"], alpha))
            return bool(score >= thresh), float(score)
        # single:304-316: the three planes' singular values in one batched call, the NC of each channel on the host
        S_cw = ctx.ref_sigma_planes(np.ascontiguousarray(np.moveaxis(stego, -1, 0)))
        nc = []
        for ch, n in enumerate("bgr"):
            Sc = np.asarray(meta["S" + n], dtype=np.float32).reshape(-1); Sw = np.asarray(meta["SW" + n], dtype=np.float32).reshape(-1)
            L = min(len(Sc), S_cw.shape[1], len(Sw))                       # single:311-313
            nc.append(_nc(Sw[:L], (S_cw[ch, :L] - Sc[:L]) / np.float32(max(alpha, 1e-8))))
        score = float((nc[0] + nc[1] + nc[2]) / 3.0)
        return bool(score >= thresh), score
    if mode == "gray":
        Y = ctx.color("bgr2y", stego)                                      # single:296
        score = float(ctx.detect_tiles(Y, meta["Sc"], meta["Sw"], alpha)[0])   # single:297-301
        return bool(score >= thresh), score                                # single:302
    planes = np.ascontiguousarray(np.moveaxis(stego, -1, 0))               # single:303
    Sc = np.stack([meta["S" + n] for n in "bgr"])
    Sw = np.stack([meta["SW" + n] for n in "bgr"])
    nc = ctx.detect_tiles(planes, Sc, Sw, alpha)                           # single:304-316
    score = float((nc[0] + nc[1] + nc[2]) / 3.0)                           # single:317
    return bool(score >= thresh), score                                    # single:318


# ---------------------------------------------------------------------------
# file level: the reference's public functions
# ---------------------------------------------------------------------------
def embed(cover_path: str, wm_source: str, out_path: str, meta_path: str,
          alpha: float = 0.1, color: bool = False, password: Optional[str] = None,
          kfrac: float = K_FRAC_DEFAULT, *, tile: Optional[int] = TILE, k_floor: int = 8,
          nonce: Optional[bytes] = None, device: int = 0, compress_meta: bool = True):
    """single:112-190.  Returns (out_path, meta_path, psnr, ssim).  ``compress_meta=False`` writes the .npz
    uncompressed (np.load - and the reference's extract / detect - read either form): the tile-mode factors are
    float noise to zlib, and compressing the 70 MB of a 4K cover costs ten times the rest of the call."""
    _check_password(password, "embed")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để nhúng.")
    cover = hg.read_image_bgr(cover_path)                                  # single:117
    wm = hg.read_image_bgr(wm_source)                                      # single:118
    if nonce is None:
        nonce = os.urandom(8)                                              # single:119
    r = embed_arrays(cover, wm, password, nonce, alpha, color, kfrac, tile, k_floor, device)
    if not out_path.lower().endswith(".png"):
        out_path = os.path.splitext(out_path)[0] + "_stego.png"            # single:148-149,178-179
    if not hg.write_png(out_path, r["stego"], 0):                          # single:150,180
        raise IOError("Ghi stego thất bại.")
    hg.save_npz(meta_path, r["meta"], compressed=compress_meta)            # single:157-166,183-189 (np.savez_compressed; members deflated concurrently)
    return out_path, meta_path, r["psnr"], r["ssim"]


def extract(stego_path: str, meta_path: str, out_path: str, password: str,
            normalize: bool = True, *, enhance: bool = False, device: int = 0) -> str:
    """single:192-282.  ``enhance=True`` applies the unsharp half of the
    reference's cosmetic post-processing (CLAHE / NL-means are OpenCV-only and
    wrapped in try/except there); default writes the extracted plane as is."""
    _check_password(password, "extract")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để giải trích.")
    img = _Later(lambda: hg.read_image_bgr(stego_path))                    # single:201, decoded while the meta is read
    data = hg.load_npz(meta_path)                                          # single:195 (all members, inflated concurrently); its errors come first, as in the reference
    st = img.result()
    wm = extract_arrays(st, data, password, normalize, device)
    if enhance:
        wm = hg.unsharp(wm, 0.25 if wm.ndim == 2 else 0.15)               # single:95,109
    if not out_path.lower().endswith(".png"):
        out_path = os.path.splitext(out_path)[0] + "_wm.png"               # single:225-226,278-279
    if not hg.write_png(out_path, wm, 1):
        raise IOError("Ghi watermark thất bại.")                           # single:229,281
    return out_path


def detect(stego_path: str, meta_path: str, thresh: float = 0.6, *, device: int = 0):
    """single:291-318.  Returns (bool, score)."""
    data = np.load(meta_path, allow_pickle=False)                          # single:292
    st = hg.read_image_bgr(stego_path)                                     # single:294
    return detect_arrays(st, data, thresh, device)


embed_watermark = embed
extract_watermark = extract
