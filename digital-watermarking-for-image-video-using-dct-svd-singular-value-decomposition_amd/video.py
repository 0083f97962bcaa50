"""Video frame loop.

The reference's video modules exist only as CPython-3.12 bytecode
(``watermark/__pycache__/video_dct_svd.cpython-312.pyc``; names recovered from
its string table: ``embed_watermark_video(host_video_path, watermark_path,
output_video_path, metadata_path, alpha, frame_interval)``,
``extract_watermark_video``, ``detect_watermark_video``).  What they show is the
*shape* of the loop: the watermark is decomposed ONCE, every
``frame_interval``-th frame's luma is embedded with it, per-frame host singular
values go to the metadata, extraction averages over the marked frames.  That
shape is built here on the tile-mode kernels, batched: one K3 launch per
watermark, one K1 launch per batch of frames with the watermark sigma shared
(``sigma_w_plane_stride = 0``).

Container: YUV4MPEG2 (``.y4m``) - uncompressed planar YUV, readable and
writable with NumPy alone (there is no OpenCV / ffmpeg in this image; the
reference uses ``cv2.VideoCapture`` / ``VideoWriter('mp4v')``).  The luma plane
IS the "Y channel" the hot path works on, so no colour conversion is involved.
Arrays of frames (``[N, H, W]`` uint8 luma) are accepted directly as well.
"""
from __future__ import annotations

import os
from typing import Iterator, Optional, Tuple

import numpy as np

from . import hostapi
from . import hostglue as hg
from . import sharding

TILE = 8


# ---------------------------------------------------------------------------
# YUV4MPEG2
# ---------------------------------------------------------------------------
_CHROMA_DIV = {"420": (2, 2), "420jpeg": (2, 2), "420mpeg2": (2, 2), "420paldv": (2, 2),
               "422": (2, 1), "444": (1, 1), "mono": (0, 0)}


class Y4M:
    """Minimal 8-bit YUV4MPEG2 reader: header fields + per-frame (Y, U, V) planes."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        head = self._f.readline()
        if not head.startswith(b"YUV4MPEG2"):
            raise ValueError(f"Không mở được video: {path}")
        self.fields = head.decode("ascii", "replace").split()[1:]
        self.W = self.H = 0
        self.chroma = "420"
        for tok in self.fields:
            if tok[0] == "W": self.W = int(tok[1:])
            elif tok[0] == "H": self.H = int(tok[1:])
            elif tok[0] == "C": self.chroma = tok[1:]
        if self.chroma not in _CHROMA_DIV:
            raise ValueError(f"unsupported Y4M chroma format C{self.chroma} (8-bit 420/422/444/mono only)")
        dx, dy = _CHROMA_DIV[self.chroma]
        self.cw, self.ch = ((self.W + dx - 1) // dx, (self.H + dy - 1) // dy) if dx else (0, 0)
        self.header_line = head

    def __iter__(self) -> Iterator[Tuple[bytes, np.ndarray, np.ndarray]]:
        ysz, csz = self.W * self.H, self.cw * self.ch
        while True:
            line = self._f.readline()
            if not line:
                return
            if not line.startswith(b"FRAME"):
                raise ValueError("corrupt Y4M stream (FRAME marker expected)")
            buf = self._f.read(ysz + 2 * csz)
            if len(buf) < ysz + 2 * csz:
                raise ValueError("truncated Y4M frame")
            y = np.frombuffer(buf, np.uint8, ysz).reshape(self.H, self.W)
            yield line, y, np.frombuffer(buf, np.uint8, 2 * csz, ysz)

    def close(self):
        self._f.close()


def write_y4m(path: str, frames_y: np.ndarray, chroma: Optional[np.ndarray] = None, fps: str = "25:1",
              chroma_tag: Optional[str] = None):
    """Write luma frames [N, H, W] (+ optional packed chroma bytes per frame)."""
    n, H, W = frames_y.shape
    tag = chroma_tag or ("mono" if chroma is None else "420")
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F{fps} Ip A1:1 C{tag}\n".encode("ascii"))
        for i in range(n):
            f.write(b"FRAME\n")
            f.write(np.ascontiguousarray(frames_y[i]).tobytes())
            if chroma is not None:
                f.write(np.ascontiguousarray(chroma[i]).tobytes())


# ---------------------------------------------------------------------------
# array level: batches of luma frames
# ---------------------------------------------------------------------------
def prepare_watermark(ctx: hostapi.Context, wm_bgr: np.ndarray, H: int, W: int, key: bytes,
                      tile: Optional[int] = TILE):
    """resize -> gray -> keyed pixel shuffle -> SVD of its DCT, once per video.
    tile=8: per-tile factors; tile=None: the reference's full-plane factors."""
    wm = hg.resize_area(wm_bgr, W, H)
    idx = hg.permutation_index(H, W, key)
    wy_s = ctx.permute_planes(hg.bgr_to_gray(wm), idx)                 # single:66-72, index pass on the device
    Uw, Sw, Vwt = ctx.svd_tiles(wy_s) if tile else ctx.ref_svd(wy_s, apply_dct=True)
    return Uw, Sw, Vwt, idx


def _k_of(tile: Optional[int], kfrac: float, k_floor: int, H: int, W: int) -> int:
    """K = max(8, int(kfrac * L)) (single:137) with L = 8 per tile or min(H, W) per plane."""
    L = tile if tile else min(H, W)
    return min(L, max(int(k_floor), int(kfrac * L)))


def embed_frames(ctx: hostapi.Context, frames_y: np.ndarray, Sw: np.ndarray, alpha: float, K: int = 8,
                 batch: int = 32, tile: Optional[int] = TILE):
    """frames_y uint8 [N, H, W] -> (stego [N, H, W], Sc); one set of launches per batch.
    tile=8: Sc [N, nby, nbx, 8] (K1); tile=None: Sc [N, min(H, W)] (batched full-plane SVDs)."""
    n, H, W = frames_y.shape
    stego = np.empty_like(frames_y)
    sc = np.empty((n, H // TILE, W // TILE, 8) if tile else (n, min(H, W)), np.float32)
    for b0 in range(0, n, batch):
        if tile:
            s, c, _ = ctx.embed_tiles(frames_y[b0:b0 + batch], Sw, alpha, K)
        else:
            s, c, _ = ctx.ref_embed_planes(frames_y[b0:b0 + batch], Sw, alpha, K)
        stego[b0:b0 + batch] = s; sc[b0:b0 + batch] = c
    return stego, sc


def extract_frames_mean(ctx: hostapi.Context, frames_y: np.ndarray, Sc: np.ndarray, Uw, Vwt, alpha: float,
                        K: int = 8, batch: int = 32, tile: Optional[int] = TILE) -> np.ndarray:
    """Mean over frames of the scrambled-watermark estimates (float32 [H, W])."""
    n, H, W = frames_y.shape
    acc = np.zeros((H, W), np.float64)
    for b0 in range(0, n, batch):
        if tile:      # the batch's estimates are added on the device: one plane per batch crosses PCIe
            acc += ctx.extract_tiles(frames_y[b0:b0 + batch], Sc[b0:b0 + batch], Uw, Vwt, alpha, K, sum_planes=True)
        else:
            w = ctx.ref_extract_planes(frames_y[b0:b0 + batch], Sc[b0:b0 + batch], Uw, Vwt, alpha, K)
            acc += w.sum(axis=0, dtype=np.float64)
    return (acc / max(n, 1)).astype(np.float32)


def detect_frames(ctx: hostapi.Context, frames_y: np.ndarray, Sc: np.ndarray, Sw: np.ndarray, alpha: float,
                  batch: int = 32, tile: Optional[int] = TILE) -> np.ndarray:
    scores = np.empty(frames_y.shape[0], np.float64)
    for b0 in range(0, frames_y.shape[0], batch):
        f = ctx.detect_tiles if tile else ctx.ref_detect_planes
        scores[b0:b0 + batch] = f(frames_y[b0:b0 + batch], Sc[b0:b0 + batch], Sw, alpha)
    return scores


def embed_frames_sharded(ctx: hostapi.Context, frames_y: np.ndarray, Sw: np.ndarray, alpha: float, K: int = 8,
                         rank: int = 0, world_size: int = 1, batch: int = 32, tile: Optional[int] = TILE):
    """This rank's share [r*N//W, (r+1)*N//W) of a batch of frames (no collective:
    the watermark sigma was broadcast beforehand, sharding.broadcast_watermark)."""
    lo, hi = sharding.frame_range(rank, world_size, frames_y.shape[0])
    stego, sc = embed_frames(ctx, frames_y[lo:hi], Sw, alpha, K, batch, tile)
    return (lo, hi), stego, sc


# ---------------------------------------------------------------------------
# file level (names of the reference's bytecode-only video module)
# ---------------------------------------------------------------------------
def _marked(n_frames: int, frame_interval: int) -> np.ndarray:
    return np.arange(0, n_frames, max(1, int(frame_interval)))


def embed_watermark_video(host_video_path: str, watermark_path: str, output_video_path: str,
                          metadata_path: str, alpha: float = 0.1, frame_interval: int = 1, *,
                          password: str = "", nonce: Optional[bytes] = None, kfrac: float = hg.K_FRAC_DEFAULT,
                          k_floor: int = 8, batch: int = 32, device: int = 0, tile: Optional[int] = TILE):
    """Embed the watermark into the luma of every ``frame_interval``-th frame of a
    .y4m video.  Returns (output_video_path, metadata_path, mean PSNR of marked frames).
    tile=8: 8x8-block formulation (fast path); tile=None: one SVD per frame like the reference's
    image embed (batched over the frames of a chunk; use a smaller ``batch``, e.g. 8)."""
    if tile not in (TILE, None):
        raise ValueError("tile must be 8 or None")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để nhúng.")
    ctx = hostapi.Context(device)
    vid = Y4M(host_video_path)
    try:
        H, W = vid.H, vid.W
        if nonce is None:
            nonce = os.urandom(8)
        key = hg.derive_key(password, nonce)
        Uw, Sw, Vwt, _ = prepare_watermark(ctx, hg.read_image_bgr(watermark_path), H, W, key, tile)
        K = _k_of(tile, kfrac, k_floor, H, W)
        sc_all, psnrs, n_frames = [], [], 0
        with open(output_video_path, "wb") as out:
            out.write(vid.header_line)
            pend = []          # (frame_line, y, chroma, marked)

            def flush():
                ys = [p[1] for p in pend if p[3]]
                if ys:
                    st, sc = embed_frames(ctx, np.stack(ys), Sw, alpha, K, batch, tile)
                    sc_all.append(sc)
                j = 0
                for line, y, chroma, marked in pend:
                    yy = y
                    if marked:
                        yy = st[j]; psnrs.append(hg.psnr(y, yy)); j += 1
                    out.write(line); out.write(yy.tobytes()); out.write(chroma.tobytes())
                pend.clear()

            for line, y, chroma in vid:
                pend.append((line, y.copy(), chroma.copy(), n_frames % max(1, frame_interval) == 0))
                n_frames += 1
                if len(pend) >= batch * max(1, frame_interval):
                    flush()
            flush()
        Sc = np.concatenate(sc_all) if sc_all else np.zeros((0, H // TILE, W // TILE, 8) if tile else (0, min(H, W)), np.float32)
        digest = hg.hmac_digest(key, [Sc, Uw, Vwt])
        # uncompressed .npz: the per-frame singular values are float noise to zlib (ratio ~1.1) and compressing them was
        # 80 % of this function's time (1.7 s for 64 frames of 1080p); np.load reads either form
        np.savez(metadata_path, mode="video_gray", payload_type="image", Sc=Sc, Uw=Uw, Vwt=Vwt, Sw=Sw,
                            shape=np.array((H, W)), alpha=float(alpha), kfrac=float(kfrac),
                            frame_interval=np.int32(frame_interval), n_frames=np.int32(n_frames),
                            tile=np.int32(tile or 0), k_floor=np.int32(k_floor),
                            nonce=np.frombuffer(nonce, dtype=np.uint8), digest=np.frombuffer(digest, dtype=np.uint8))
        return output_video_path, metadata_path, float(np.mean(psnrs)) if psnrs else 99.0
    finally:
        vid.close(); ctx.close()


def _load_video_meta(metadata_path: str):
    data = np.load(metadata_path, allow_pickle=False)
    if str(data["mode"]) != "video_gray":
        raise ValueError("metadata was not written by embed_watermark_video")
    return data


def _meta_tile(data) -> Optional[int]:
    return int(data["tile"]) or None


def _marked_luma(stego_video_path: str, data) -> np.ndarray:
    vid = Y4M(stego_video_path)
    try:
        fi = int(data["frame_interval"])
        ys = [y.copy() for i, (_, y, _) in enumerate(vid) if i % max(1, fi) == 0]
    finally:
        vid.close()
    n = data["Sc"].shape[0]
    if len(ys) < n:
        raise ValueError("video has fewer marked frames than the metadata")
    return np.stack(ys[:n]) if n else np.zeros((0,) + tuple(map(int, data["shape"])), np.uint8)


def extract_watermark_video(stego_video_path: str, metadata_path: str, output_image_path: str,
                            password: str, normalize: bool = True, *, batch: int = 32, device: int = 0) -> str:
    """Averaged multi-frame extraction -> watermark image (PNG)."""
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để giải trích.")
    data = _load_video_meta(metadata_path)
    H, W = map(int, data["shape"])
    nonce = bytes(bytearray(data["nonce"].astype(np.uint8).tolist()))
    key = hg.derive_key(password, nonce)
    if not hg.digests_equal(hg.hmac_digest(key, [data["Sc"], data["Uw"], data["Vwt"]]),
                            bytes(bytearray(data["digest"].astype(np.uint8).tolist()))):
        raise ValueError("Sai mật khẩu hoặc meta không khớp.")
    ctx = hostapi.Context(device)
    try:
        ys = _marked_luma(stego_video_path, data)
        tile = _meta_tile(data)
        K = _k_of(tile, float(data["kfrac"]), int(data["k_floor"]), H, W)
        wy_s = extract_frames_mean(ctx, ys, data["Sc"], data["Uw"], data["Vwt"], float(data["alpha"]), K, batch, tile)
        img = ctx.unpermute_normalize_u8(wy_s, hg.permutation_index(H, W, key), normalize)   # single:74-80, 221-222 on the device
    finally:
        ctx.close()
    if not output_image_path.lower().endswith(".png"):
        output_image_path = os.path.splitext(output_image_path)[0] + "_wm.png"
    if not hg.write_png(output_image_path, img, 1):
        raise IOError("Ghi watermark thất bại.")
    return output_image_path


def detect_watermark_video(stego_video_path: str, metadata_path: str, thresh: float = 0.6, *,
                           batch: int = 32, device: int = 0):
    """(bool, mean score, per-frame scores) over the marked frames."""
    data = _load_video_meta(metadata_path)
    ctx = hostapi.Context(device)
    try:
        ys = _marked_luma(stego_video_path, data)
        scores = detect_frames(ctx, ys, data["Sc"], data["Sw"], float(data["alpha"]), batch, _meta_tile(data))
    finally:
        ctx.close()
    mean = float(scores.mean()) if scores.size else 0.0
    return bool(mean >= thresh), mean, scores


# ---------------------------------------------------------------------------
# colour video (names of the reference's bytecode-only ``color_video_dct_svd`` module: ``embed_watermark_video_color`` /
# ``extract_watermark_video_color``; its source is not in the tree, so the SHAPE is the colour image embed of single:121-166
# put into the luma loop above: the colour watermark's B, G, R planes are decomposed ONCE, every marked frame's B, G, R planes
# are embedded with them, per-frame host singular values per channel go to the metadata, extraction averages over the marked
# frames per channel).  Container: 8-bit 4:4:4 ``.y4m`` (planes Y, Cb, Cr); the frames pass through OpenCV's fixed-point
# YCrCb <-> BGR conversion on the device (``wm_color_u8``) on the way in and out, so - like any YUV container - the stored
# stego differs from the embedded BGR planes by that conversion's rounding (a grey level or two per channel).  No audio remux.
# ---------------------------------------------------------------------------
def _frame_to_bgr_planes(ctx: hostapi.Context, y: np.ndarray, chroma: np.ndarray, H: int, W: int) -> np.ndarray:
    cb = chroma[:H * W].reshape(H, W); cr = chroma[H * W:].reshape(H, W)
    bgr = ctx.color("ycrcb2bgr", np.ascontiguousarray(np.stack([y, cr, cb], axis=-1)))       # OpenCV order: Y, Cr, Cb
    return np.ascontiguousarray(np.moveaxis(bgr, -1, 0))                                       # [3, H, W]: B, G, R


def _bgr_planes_to_frame(ctx: hostapi.Context, planes: np.ndarray):
    ycc = ctx.color("bgr2ycrcb", np.ascontiguousarray(np.moveaxis(planes, 0, -1)))
    return np.ascontiguousarray(ycc[..., 0]), np.ascontiguousarray(ycc[..., 2]), np.ascontiguousarray(ycc[..., 1])   # Y, Cb, Cr


def prepare_watermark_color(ctx: hostapi.Context, wm_bgr: np.ndarray, H: int, W: int, key: bytes, tile: Optional[int] = TILE):
    """resize -> B, G, R planes -> ONE keyed pixel shuffle for all three (single:124-126) -> their three decompositions in one
    batched call.  Returns (Uw [3, ...], Sw [3, ...], Vwt [3, ...], idx)."""
    wm = hg.resize_area(wm_bgr, W, H)
    idx = hg.permutation_index(H, W, key)
    planes = ctx.permute_planes(np.ascontiguousarray(np.moveaxis(wm, -1, 0)), idx)
    if tile:
        Uw, Sw, Vwt = ctx.svd_tiles(planes)
    else:
        Uw, Sw, Vwt = ctx.ref_svd_planes(planes.astype(np.float32), apply_dct=True)
    return Uw, Sw, Vwt, idx


_CH = "bgr"


def embed_frames_color(ctx: hostapi.Context, planes: np.ndarray, Sw: np.ndarray, alpha: float, K: int = 8, batch: int = 8,
                       tile: Optional[int] = TILE):
    """planes uint8 [n, 3, H, W] (B, G, R of n frames), Sw [3, ...] -> (stego [n, 3, H, W], [Sb, Sg, Sr]): channel c of every
    frame gets the watermark's channel c (single:139-152); one set of launches per channel and batch."""
    st = np.empty_like(planes)
    sc = []
    for ch in range(3):
        s, c = embed_frames(ctx, np.ascontiguousarray(planes[:, ch]), Sw[ch], alpha, K, batch, tile)
        st[:, ch] = s; sc.append(c)
    return st, sc


def embed_watermark_video_color(host_video_path: str, watermark_path: str, output_video_path: str,
                                metadata_path: str, alpha: float = 0.1, frame_interval: int = 1, *,
                                password: str = "", nonce: Optional[bytes] = None, kfrac: float = hg.K_FRAC_DEFAULT,
                                k_floor: int = 8, batch: int = 8, device: int = 0, tile: Optional[int] = TILE):
    """Embed a colour watermark into the B, G, R planes of every ``frame_interval``-th frame of a 4:4:4 .y4m video.
    Returns (output_video_path, metadata_path, mean PSNR of the marked frames' BGR planes)."""
    if tile not in (TILE, None):
        raise ValueError("tile must be 8 or None")
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để nhúng.")
    ctx = hostapi.Context(device)
    vid = Y4M(host_video_path)
    try:
        if vid.chroma != "444":
            raise ValueError("embed_watermark_video_color needs an 8-bit 4:4:4 .y4m (C444): per-channel embedding needs full-resolution chroma")
        H, W = vid.H, vid.W
        if nonce is None:
            nonce = os.urandom(8)
        key = hg.derive_key(password, nonce)
        Uw, Sw, Vwt, _ = prepare_watermark_color(ctx, hg.read_image_bgr(watermark_path), H, W, key, tile)
        K = _k_of(tile, kfrac, k_floor, H, W)
        sc_all = [[], [], []]
        psnrs, n_frames = [], 0
        with open(output_video_path, "wb") as out:
            out.write(vid.header_line)
            pend = []          # (frame_line, y, chroma, marked)

            def flush():
                marked = [p for p in pend if p[3]]
                if marked:
                    planes = np.stack([_frame_to_bgr_planes(ctx, p[1], p[2], H, W) for p in marked])     # [n, 3, H, W]
                    st, sc = embed_frames_color(ctx, planes, Sw, alpha, K, batch, tile)
                    for ch in range(3):
                        sc_all[ch].append(sc[ch])
                    psnrs.extend(hg.psnr(planes[i], st[i]) for i in range(len(marked)))
                j = 0
                for line, y, chroma, is_marked in pend:
                    if is_marked:
                        yy, cb, cr = _bgr_planes_to_frame(ctx, st[j]); j += 1
                        out.write(line); out.write(yy.tobytes()); out.write(cb.tobytes()); out.write(cr.tobytes())
                    else:
                        out.write(line); out.write(y.tobytes()); out.write(chroma.tobytes())
                pend.clear()

            for line, y, chroma in vid:
                pend.append((line, y.copy(), chroma.copy(), n_frames % max(1, frame_interval) == 0))
                n_frames += 1
                if len(pend) >= batch * max(1, frame_interval):
                    flush()
            flush()
        empty = np.zeros((0, H // TILE, W // TILE, 8) if tile else (0, min(H, W)), np.float32)
        S = [np.concatenate(sc_all[ch]) if sc_all[ch] else empty for ch in range(3)]
        digest = hg.hmac_digest(key, S + [Uw[ch] for ch in range(3)] + [Vwt[ch] for ch in range(3)])     # the coverage of single:160-161
        meta = dict(mode="video_color", payload_type="image", shape=np.array((H, W)), alpha=float(alpha), kfrac=float(kfrac),
                    frame_interval=np.int32(frame_interval), n_frames=np.int32(n_frames), tile=np.int32(tile or 0),
                    k_floor=np.int32(k_floor), nonce=np.frombuffer(nonce, dtype=np.uint8), digest=np.frombuffer(digest, dtype=np.uint8))
        for ch, n in enumerate(_CH):                     # the colour image meta's key names (single:157-166)
            meta["S" + n] = S[ch]; meta["UW" + n] = Uw[ch]; meta["VW" + n + "t"] = Vwt[ch]; meta["SW" + n] = Sw[ch]
        np.savez(metadata_path, **meta)
        return output_video_path, metadata_path, float(np.mean(psnrs)) if psnrs else 99.0
    finally:
        vid.close(); ctx.close()


def _load_video_meta_color(metadata_path: str):
    data = np.load(metadata_path, allow_pickle=False)
    if str(data["mode"]) != "video_color":
        raise ValueError("metadata was not written by embed_watermark_video_color")
    return data


def _marked_bgr(ctx: hostapi.Context, stego_video_path: str, data) -> np.ndarray:
    vid = Y4M(stego_video_path)
    try:
        if vid.chroma != "444":
            raise ValueError("a colour-watermarked video is 4:4:4")
        fi = int(data["frame_interval"]); n = data["Sb"].shape[0]
        H, W = vid.H, vid.W
        out = []
        for i, (_, y, chroma) in enumerate(vid):
            if i % max(1, fi) == 0 and len(out) < n:
                out.append(_frame_to_bgr_planes(ctx, y, chroma, H, W))
    finally:
        vid.close()
    if len(out) < n:
        raise ValueError("video has fewer marked frames than the metadata")
    return np.stack(out) if out else np.zeros((0, 3) + tuple(map(int, data["shape"])), np.uint8)


def extract_watermark_video_color(stego_video_path: str, metadata_path: str, output_image_path: str, password: str,
                                  normalize: bool = True, *, batch: int = 8, device: int = 0) -> str:
    """Averaged multi-frame extraction per channel -> colour watermark image (PNG)."""
    if not password:
        raise ValueError("Vui lòng nhập mật khẩu để giải trích.")
    data = _load_video_meta_color(metadata_path)
    H, W = map(int, data["shape"])
    nonce = bytes(bytearray(data["nonce"].astype(np.uint8).tolist()))
    key = hg.derive_key(password, nonce)
    parts = [data["S" + n] for n in _CH] + [data["UW" + n] for n in _CH] + [data["VW" + n + "t"] for n in _CH]
    if not hg.digests_equal(hg.hmac_digest(key, parts), bytes(bytearray(data["digest"].astype(np.uint8).tolist()))):
        raise ValueError("Sai mật khẩu hoặc meta không khớp.")
    ctx = hostapi.Context(device)
    try:
        planes = _marked_bgr(ctx, stego_video_path, data)
        tile = _meta_tile(data)
        K = _k_of(tile, float(data["kfrac"]), int(data["k_floor"]), H, W)
        idx = hg.permutation_index(H, W, key)
        chans = []
        for ch, n in enumerate(_CH):
            w_s = extract_frames_mean(ctx, np.ascontiguousarray(planes[:, ch]), data["S" + n], data["UW" + n], data["VW" + n + "t"],
                                      float(data["alpha"]), K, batch, tile)
            chans.append(ctx.unpermute_normalize_u8(w_s, idx, normalize))                    # single:265-271 per channel
        img = np.stack(chans, axis=-1)
    finally:
        ctx.close()
    if not output_image_path.lower().endswith(".png"):
        output_image_path = os.path.splitext(output_image_path)[0] + "_wm.png"
    if not hg.write_png(output_image_path, img, 1):
        raise IOError("Ghi watermark thất bại.")
    return output_image_path


def detect_watermark_video_color(stego_video_path: str, metadata_path: str, thresh: float = 0.6, *,
                                 batch: int = 8, device: int = 0):
    """(bool, mean score, per-frame scores): a frame's score is the mean of its three channels' NC (single:317)."""
    data = _load_video_meta_color(metadata_path)
    ctx = hostapi.Context(device)
    try:
        planes = _marked_bgr(ctx, stego_video_path, data)
        tile = _meta_tile(data)
        per_ch = [detect_frames(ctx, np.ascontiguousarray(planes[:, ch]), data["S" + n], data["SW" + n], float(data["alpha"]), batch, tile)
                  for ch, n in enumerate(_CH)]
    finally:
        ctx.close()
    scores = (per_ch[0] + per_ch[1] + per_ch[2]) / 3.0
    mean = float(scores.mean()) if scores.size else 0.0
    return bool(mean >= thresh), mean, scores
