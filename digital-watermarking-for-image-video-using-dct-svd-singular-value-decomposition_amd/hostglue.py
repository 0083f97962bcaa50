"""Host-side glue of the drop-in (NumPy / SciPy / Pillow): the parts of the
reference that BASELINE.json's north_star keeps on the host - image I/O,
colour conversion, watermark resize, password -> key -> permutation, HMAC,
PSNR / SSIM.  Citations are into /root/reference/app_dct_svd_single.py
("single").  OpenCV is not available in this image, so every ``cv2.*`` call
is restated from OpenCV's documented formulas.  No numerics of the hot path
(DCT / SVD / reconstruction) live here - those are HIP kernels behind
include/wmhip.h.
"""
from __future__ import annotations

import hashlib
import time
import os
import hmac as _hmac
import threading
from collections import OrderedDict

import numpy as np
import scipy.ndimage
from PIL import Image

K_FRAC_DEFAULT = 0.6  # single:13


# ---- image I/O  (single:15-19, 150, 180, 228, 280) --------------------------
def _read_png_unfiltered(path: str):
    """Fast path for the PNGs this module writes itself (8-bit gray / RGB, not interlaced, every scanline with filter type
    0): inflate and reshape - no per-pixel unfiltering, a 4K stego in 15 ms instead of 60.  Anything else (other colour
    types, bit depths, interlacing, a palette, any filtered scanline, a damaged stream) returns None and goes to Pillow,
    whose own checks then decide.  The same guards as Pillow's apply here: every chunk's CRC is verified, an image over
    ``Image.MAX_IMAGE_PIXELS`` is refused, and the stream is inflated with a bound of exactly the bytes the header
    announces (+ 1, to notice a longer stream) - a small file cannot be made to allocate more than its header's size."""
    import struct
    import zlib
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:8] != b"\x89PNG\r\n\x1a\n":
            return None
        pos, idat, ihdr = 8, [], None
        while pos + 12 <= len(data):
            (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
            if pos + 12 + n > len(data):
                return None                                           # truncated chunk
            body = data[pos + 8:pos + 8 + n]
            (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
            if zlib.crc32(body, zlib.crc32(tag)) & 0xFFFFFFFF != crc:
                return None                                           # damaged: let Pillow report it
            if tag == b"IHDR":
                if ihdr is not None or n != 13:
                    return None
                ihdr = struct.unpack(">IIBBBBB", body)
            elif tag == b"IDAT":
                idat.append(body)
            elif tag == b"IEND":
                break
            elif tag in (b"PLTE", b"tRNS") or not (tag[0] & 0x20):      # palette / transparency, or a critical chunk this reader does not know
                return None
            pos += 12 + n
        if ihdr is None or not idat:
            return None
        w, h, depth, ctype, _, _, interlace = ihdr
        if depth != 8 or ctype not in (0, 2) or interlace != 0 or w == 0 or h == 0:
            return None
        limit = Image.MAX_IMAGE_PIXELS
        if limit is not None and w * h > limit:
            return None                                               # Pillow's decompression-bomb guard decides
        ch = 1 if ctype == 0 else 3
        expect = h * (1 + w * ch)
        d = zlib.decompressobj()
        raw = d.decompress(b"".join(idat), expect + 1)                # never materialises more than the header's size
        if len(raw) != expect or d.unconsumed_tail or not d.eof:
            return None
        a = np.frombuffer(raw, np.uint8).reshape(h, 1 + w * ch)
        if a[:, 0].any():
            return None
        px = a[:, 1:].reshape(h, w, ch)
        return np.ascontiguousarray(px[..., ::-1]) if ch == 3 else np.repeat(px, 3, axis=2)
    except Exception:
        return None


def read_image_bgr(path: str) -> np.ndarray:
    """``cv2.imread(path, cv2.IMREAD_COLOR)``: always 3-channel BGR uint8."""
    fast = _read_png_unfiltered(path) if str(path).lower().endswith(".png") else None
    if fast is not None:
        return fast
    try:
        with Image.open(path) as im:
            if im.mode in ("I;16", "I;16L", "I;16B", "I"):         # 16-bit gray: OpenCV keeps the high byte
                g = (np.asarray(im).astype(np.uint32) >> 8).clip(0, 255).astype(np.uint8)
                rgb = np.repeat(g[..., None], 3, axis=2)
            else:
                rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception:
        raise ValueError(f"Không mở được ảnh: {path}")          # single:17-18
    return np.ascontiguousarray(rgb[..., ::-1])


_DEFLATE_CHUNK = 1 << 20


def _deflate_chunks(parts, level: int = 6, threads: int = 32):
    """Raw-deflate streams of several byte buffers, every buffer cut into 1 MiB chunks that are deflated from a fresh state
    on worker threads (zlib releases the GIL) and closed with a sync flush - byte-aligned, not final -, the last one with
    Z_FINISH: concatenated, a buffer's chunks are ONE valid raw-deflate stream (pigz's construction).  parts: list of
    (prefix bytes, uint8 array); returns the list of compressed streams in the same order."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for i, (_, data) in enumerate(parts):
        n_chunks = max(1, -(-data.size // _DEFLATE_CHUNK))
        jobs += [(i, c, n_chunks) for c in range(n_chunks)]

    def deflate(job):
        i, c, n_chunks = job
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        out = co.compress(parts[i][0]) if c == 0 and parts[i][0] else b""
        out += co.compress(parts[i][1][c * _DEFLATE_CHUNK:(c + 1) * _DEFLATE_CHUNK])
        return out + (co.flush(zlib.Z_FINISH) if c == n_chunks - 1 else co.flush(zlib.Z_SYNC_FLUSH))

    if len(jobs) == 1:
        done = [deflate(jobs[0])]
    else:
        with ThreadPoolExecutor(max_workers=max(1, min(threads, len(jobs), os.cpu_count() or 1))) as ex:
            done = list(ex.map(deflate, jobs))
    return [b"".join(d for (j, _, _), d in zip(jobs, done) if j == i) for i in range(len(parts))]


def _png_bytes(img: np.ndarray, compression: int) -> bytes:
    """An 8-bit gray or RGB PNG written directly: scanlines with filter type 0, ONE zlib stream at the requested level
    (level 0 - the reference's stego files - is stored blocks at memcpy speed), CRC per chunk.  Pillow's encoder took
    60-100 ms for a 1080p stego at level 0 (it still walks the filter heuristics); this takes a tenth of that."""
    import struct
    import zlib
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else 3
    raw = np.empty((h, 1 + w * ch), np.uint8)
    raw[:, 0] = 0                                                   # filter type 0 (None) on every scanline
    raw[:, 1:] = img.reshape(h, w * ch)
    level = int(min(max(compression, 0), 9))
    if level == 0 or raw.size <= _DEFLATE_CHUNK:
        comp = zlib.compress(raw, level)
    else:       # an extracted 1080p colour watermark is 6 MB of noise: 100 ms through one zlib.compress(level 1), a tenth in chunks
        flat = raw.reshape(-1)
        comp = b"\x78\x01" + _deflate_chunks([(b"", flat)], level)[0] + struct.pack(">I", zlib.adler32(flat) & 0xFFFFFFFF)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if ch == 1 else 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", comp) + chunk(b"IEND", b"")


def write_png(path: str, img: np.ndarray, compression: int = 0) -> bool:
    """``cv2.imwrite(path, img, [IMWRITE_PNG_COMPRESSION, c])`` for BGR or gray uint8 (same pixels; the byte stream is this
    module's own - filter type 0 throughout - where libpng picks filters per scanline)."""
    try:
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3) or img.size == 0 \
                or img.shape[0] * (1 + img.shape[1] * (1 if img.ndim == 2 else 3)) >= (1 << 31):
            if img.ndim == 2:                                       # anything unusual: Pillow's encoder
                Image.fromarray(img, mode="L").save(path, format="PNG", compress_level=compression)
            else:
                Image.fromarray(np.ascontiguousarray(img[..., ::-1]), mode="RGB").save(path, format="PNG", compress_level=compression)
            return True
        data = _png_bytes(img if img.ndim == 2 else np.ascontiguousarray(img[..., ::-1]), compression)
        with open(path, "wb") as f:
            f.write(data)
        return True
    except Exception:
        return False


# ---- colour conversion (OpenCV 8-bit fixed point) ---------------------------
def bgr_to_gray(bgr: np.ndarray) -> np.ndarray:
    """cv2.COLOR_BGR2GRAY: (B*3735 + G*19235 + R*9798 + 2^14) >> 15."""
    b = bgr[..., 0].astype(np.int32); g = bgr[..., 1].astype(np.int32); r = bgr[..., 2].astype(np.int32)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def bgr_to_ycrcb(bgr: np.ndarray) -> np.ndarray:
    """cv2.COLOR_BGR2YCrCb (single:22), 14-bit coefficients 4899/9617/1868, 11682, 9241."""
    b = bgr[..., 0].astype(np.int32); g = bgr[..., 1].astype(np.int32); r = bgr[..., 2].astype(np.int32)
    y = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
    cr = ((r - y) * 11682 + (128 << 14) + 8192) >> 14
    cb = ((b - y) * 9241 + (128 << 14) + 8192) >> 14
    out = np.stack([y, cr, cb], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def ycrcb_to_bgr(ycc: np.ndarray) -> np.ndarray:
    """cv2.COLOR_YCrCb2BGR (single:30), coefficients 22987, -11698, -5636, 29049."""
    y = ycc[..., 0].astype(np.int32)
    cr = ycc[..., 1].astype(np.int32) - 128
    cb = ycc[..., 2].astype(np.int32) - 128
    b = y + ((cb * 29049 + 8192) >> 14)
    g = y + ((cb * -5636 + cr * -11698 + 8192) >> 14)
    r = y + ((cr * 22987 + 8192) >> 14)
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)


# ---- watermark resize (single:118) -------------------------------------------
def _area_matrix(n_src: int, n_dst: int, box: bool) -> np.ndarray:
    M = np.zeros((n_dst, n_src), np.float64)
    scale = n_src / n_dst
    if box:                    # shrink: fractional box coverage
        for d in range(n_dst):
            lo, hi = d * scale, (d + 1) * scale
            for s_ in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_src)):
                M[d, s_] = max(0.0, min(hi, s_ + 1) - max(lo, s_))
            M[d] /= M[d].sum()
    else:                      # INTER_AREA's linear variant (sx = floor(dx*scale), area-style fx)
        inv = 1.0 / scale
        for d in range(n_dst):
            s_ = int(np.floor(d * scale))
            fx = (d + 1) - (s_ + 1) * inv
            fx = 0.0 if fx <= 0 else fx - np.floor(fx)
            M[d, s_] += 1.0 - fx
            M[d, min(s_ + 1, n_src - 1)] += fx
    return M


def resize_area(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """``cv2.resize(img, (W, H), interpolation=cv2.INTER_AREA)`` on uint8.
    Integer enlargement is pixel replication, integer reduction a box mean.  cv::resize takes
    the box filter only when BOTH axes shrink; as soon as one axis enlarges, both axes go
    through its linear variant (chosen jointly, not per axis).  Parity with OpenCV is unpinned
    (cv2 is not importable here): OpenCV's 8-bit paths round with cvRound (half to even) after
    the box filter and use 11-bit fixed-point coefficients in the linear variant; this
    restatement uses exact weights and round-half-up."""
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img.copy()
    if H % h == 0 and W % w == 0:
        return np.repeat(np.repeat(img, H // h, axis=0), W // w, axis=1)
    box = h >= H and w >= W
    My, Mx = _area_matrix(h, H, box), _area_matrix(w, W, box)
    src = img.astype(np.float64)
    planes = [src] if src.ndim == 2 else [src[..., c] for c in range(src.shape[2])]
    # every channel is resized like a gray image: two BLAS products, evaluated left to right like the oracle's
    # (float64 products associate differently otherwise, and box-filter results sit on exact .5 ties often enough
    # for the last bit to show); a 3-operand einsum walks H*W*h*w*c terms - 70 s for a 64x64 logo on a 1080p cover.
    # Rounding is half up: floor(x + 0.5), which for the non-negative values of a convex combination of uint8
    # pixels is what the float64 -> uint8 cast of x + 0.5 does (truncation), so the passes over the H x W
    # doubles (0.17 s of a 0.23 s embed_arrays at 4K in round 1's five-pass form) are one in-place add and one
    # narrowing copy into a planar buffer, interleaved at the end.
    # One float64 H x W buffer for all channels (fresh 66 MB arrays per channel were mostly page faults) and the
    # narrowing store goes straight into the interleaved result - the products and their rounding are unchanged.
    out = np.empty((H, W) if src.ndim == 2 else (H, W, len(planes)), np.uint8)
    MxT = Mx.T          # the VIEW: a contiguous copy takes another BLAS path and moves box-filter results across .5 ties
    t_ = np.empty((H, W), np.float64)
    for c, p_ in enumerate(planes):
        np.matmul(My @ p_, MxT, out=t_)
        np.add(t_, 0.5, out=t_)
        if src.ndim == 2:
            out[...] = t_                           # assignment truncates like astype (same_kind casting would refuse)
        else:
            out[..., c] = t_
    return out


_resize_cache: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
_resize_lock = threading.Lock()
_RESIZE_CACHE_ENTRIES = 2


def resize_area_cached(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """resize_area, memoised on the image's content (two entries): a caller that embeds one logo into many covers
    of one size - the reference's GUI and batch use - pays the two float64 products (0.18 s of a 0.24 s 4K
    embed_arrays) once.  The result is shared and marked read-only."""
    key = (hashlib.sha1(np.ascontiguousarray(img).tobytes()).digest(), img.shape, str(img.dtype), int(W), int(H))
    with _resize_lock:
        hit = _resize_cache.get(key)
        if hit is not None:
            _resize_cache.move_to_end(key)
            return hit
    out = resize_area(img, W, H)
    out.setflags(write=False)
    with _resize_lock:
        _resize_cache[key] = out
        while len(_resize_cache) > _RESIZE_CACHE_ENTRIES:
            _resize_cache.popitem(last=False)
    return out


# ---- security wrapper (single:59-86) -----------------------------------------
def derive_key(password: str, nonce: bytes) -> bytes:
    return hashlib.sha256(password.encode("utf-8") + nonce).digest()


def rng_from_key(key: bytes) -> np.random.Generator:
    return np.random.default_rng(int.from_bytes(key[:8], "big", signed=False))


class KeyedIndex(np.ndarray):
    """The permutation of one (H, W, key), carrying that identity as ``tag`` so that a device-side copy can be cached
    on WHAT the index is rather than on where it lives (NumPy hands a freed index's address to the next arange of the
    same size).  Views and copies made from it do not inherit the tag."""
    tag = None

    def __array_finalize__(self, obj):
        self.tag = None


_perm_cache: "OrderedDict[tuple, np.ndarray]" = OrderedDict()
_perm_lock = threading.Lock()
_PERM_CACHE_ENTRIES = 2


def permutation_is_cached(H: int, W: int, key: bytes) -> bool:
    """Whether permutation_index(H, W, key) would be served from the cache (an authenticated key's index usually is)."""
    with _perm_lock:
        return (int(H), int(W), bytes(key)) in _perm_cache


def permutation_index(H: int, W: int, key: bytes) -> np.ndarray:
    """``idx = np.arange(H*W); rng.shuffle(idx)`` (single:68-69,124,219,265):
    bit-exact because it *is* the same NumPy PCG64 call.  The shuffle is sequential host work
    (0.15 s for a 4K plane), so the last two (H, W, key) results are kept: an extract right after an
    embed, or the frames of one video, pay for it once.  The returned array is read-only."""
    k = (int(H), int(W), bytes(key))
    with _perm_lock:
        idx = _perm_cache.get(k)
        if idx is not None:
            _perm_cache.move_to_end(k)
            return idx
    idx = np.arange(H * W)
    rng_from_key(key).shuffle(idx)
    idx = idx.view(KeyedIndex)
    idx.tag = ("perm", int(H), int(W), hashlib.sha256(bytes(key)).digest())
    idx.setflags(write=False)
    with _perm_lock:
        _perm_cache[k] = idx
        while len(_perm_cache) > _PERM_CACHE_ENTRIES:
            _perm_cache.popitem(last=False)
    return idx


def permute(plane: np.ndarray, idx: np.ndarray) -> np.ndarray:
    H, W = plane.shape[:2]
    return plane.reshape(-1)[idx].reshape(H, W).astype(np.float32)


def unpermute(plane: np.ndarray, idx: np.ndarray) -> np.ndarray:
    H, W = plane.shape[:2]
    inv = np.empty_like(idx)
    inv[idx] = np.arange(idx.size)
    return plane.reshape(-1)[inv].reshape(H, W)


def save_npz(path: str, arrays: dict, compressed: bool = True, threads: int = 32) -> str:
    """``np.savez_compressed(path, **arrays)`` (single:157-166,183-189) with the members deflated CONCURRENTLY: the factors of
    a tile-mode or colour meta are a handful of large float arrays, zlib releases the GIL, and a .npz is a plain zip - so
    each member is serialised (the same .npy 1.0 header NumPy writes), deflated in chunks and CRC'd on worker threads and the
    container is written by hand (stored sizes < 4 GiB; anything larger, or ``compressed=False``, goes through NumPy).
    ``np.load`` - the reference's reader - sees an ordinary compressed .npz.  Returns the path written (NumPy's rule: '.npz' is
    appended when missing)."""
    import io
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    path = os.fspath(path)                          # np.savez accepts os.PathLike too
    if not path.endswith(".npz"):
        path = path + ".npz"
    items = [(k, np.asanyarray(v)) for k, v in arrays.items()]
    if not compressed or any(a.dtype.hasobject for _, a in items) or sum(a.nbytes for _, a in items) >= (1 << 31):
        (np.savez_compressed if compressed else np.savez)(path, **arrays)
        return path

    # every member's deflate stream in 1 MiB chunks on worker threads (_deflate_chunks), its CRC beside them
    heads, datas = [], []
    for name, a in items:
        # the header describes the array whose bytes are written: the C-contiguous copy (a Fortran-ordered or transposed
        # member would otherwise be announced as fortran_order=True over C-ordered bytes and load scrambled)
        c = np.ascontiguousarray(a)
        if a.ndim == 0:
            c = c.reshape(())                       # ascontiguousarray promotes 0-d to 1-d
        head = io.BytesIO()
        np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(c))
        heads.append(head.getvalue())
        datas.append(c.reshape(-1).view(np.uint8) if c.size else np.zeros(0, np.uint8))
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(items)))) as ex:
        crcs = [ex.submit(lambda i=i: zlib.crc32(datas[i], zlib.crc32(heads[i])) & 0xFFFFFFFF) for i in range(len(items))]
        comps = _deflate_chunks(list(zip(heads, datas)), 6, threads)
        members = [((name + ".npy").encode("utf-8"), comps[i], crcs[i].result(), len(heads[i]) + datas[i].size)
                   for i, (name, _) in enumerate(items)]
    t = time.localtime()
    dostime = (t.tm_hour << 11) | (t.tm_min << 5) | (t.tm_sec // 2)
    dosdate = ((max(t.tm_year, 1980) - 1980) << 9) | (t.tm_mon << 5) | t.tm_mday
    central = []
    with open(path, "wb") as f:
        for name, comp, crc, usize in members:
            off = f.tell()
            f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, 8, dostime, dosdate, crc, len(comp), usize, len(name), 0))
            f.write(name); f.write(comp)
            central.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 20, 20, 0, 8, dostime, dosdate, crc, len(comp), usize,
                                       len(name), 0, 0, 0, 0, 0o600 << 16, off) + name)
        cd_off = f.tell()
        for c in central:
            f.write(c)
        cd_size = f.tell() - cd_off
        f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(central), len(central), cd_size, cd_off, 0))
    return path


def load_npz(path: str, threads: int = 8) -> dict:
    """Every member of a .npz (np.load(path, allow_pickle=False), single:195) read CONCURRENTLY into a dict: extract needs
    all the factors, a colour meta holds six large ones, and inflating them one after the other was most of a file-level
    extract (zlib releases the GIL; every worker opens its own handle).  KeyError / ValueError behaviour of the
    members is NumPy's."""
    from concurrent.futures import ThreadPoolExecutor
    with np.load(path, allow_pickle=False) as z:
        names = list(z.files)

    def rd(k):
        with np.load(path, allow_pickle=False) as z:
            return k, z[k]
    if len(names) <= 1:
        return dict(rd(k) for k in names)
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(names)))) as ex:
        return dict(ex.map(rd, names))


def hmac_digest(key: bytes, arrays) -> bytes:
    h = _hmac.new(key, b"", hashlib.sha256)
    for a in arrays:
        # the array's own bytes (the reference's `.tobytes()`, single:82-86) fed as a buffer: no 33 MB copy per factor at 4K
        h.update(np.ascontiguousarray(a).reshape(-1).view(np.uint8))
    return h.digest()


def digests_equal(a: bytes, b: bytes) -> bool:
    return _hmac.compare_digest(a, b)


# ---- metrics (single:38-57) ----------------------------------------------------
def psnr(a: np.ndarray, b: np.ndarray) -> float:
    a = a.astype(np.float32); b = b.astype(np.float32)
    mse = float(np.mean((a - b) ** 2))
    if mse <= 1e-12:
        return 99.0
    return float(20.0 * np.log10(255.0 / max(np.sqrt(mse), 1e-12)))


_G11 = None


def _blur(img: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(img, (11, 11), 1.5), default border REFLECT_101."""
    global _G11
    if _G11 is None:
        x = np.arange(11, dtype=np.float64) - 5.0
        k = np.exp(-(x * x) / (2 * 1.5 * 1.5))
        _G11 = (k / k.sum()).astype(np.float32)
    t = scipy.ndimage.correlate1d(img, _G11, axis=0, mode="mirror")
    return scipy.ndimage.correlate1d(t, _G11, axis=1, mode="mirror")


def ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    if img1.ndim == 3: img1 = bgr_to_gray(img1)
    if img2.ndim == 3: img2 = bgr_to_gray(img2)
    x = img1.astype(np.float32); y = img2.astype(np.float32)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mx, my = _blur(x), _blur(y)
    sxx = _blur(x * x) - mx * mx
    syy = _blur(y * y) - my * my
    sxy = _blur(x * y) - mx * my
    num = (2 * mx * my + C1) * (2 * sxy + C2)
    den = (mx * mx + my * my + C1) * (sxx + syy + C2) + 1e-12
    return float(np.mean(num / den))


def normalize_minmax(x: np.ndarray) -> np.ndarray:
    """``cv2.normalize(x, None, 0, 255, cv2.NORM_MINMAX)`` (single:221,269-271)."""
    x = x.astype(np.float32)
    lo, hi = float(x.min()), float(x.max())
    scale = 255.0 / (hi - lo) if (hi - lo) > np.finfo(np.float64).eps else 0.0
    return ((x - np.float32(lo)) * np.float32(scale)).astype(np.float32)


def unsharp(img_u8: np.ndarray, amount: float) -> np.ndarray:
    """The unsharp half of ``_enhance_gray/_enhance_color`` (single:94-96,108-110):
    GaussianBlur(sigma=1.0) then addWeighted(e, 1+a, blur, -a).  CLAHE and
    NL-means (OpenCV-only, inside try/except in the reference) are not applied."""
    f = img_u8.astype(np.float32)
    sig = (1.0, 1.0) if f.ndim == 2 else (1.0, 1.0, 0.0)
    blur = scipy.ndimage.gaussian_filter(f, sigma=sig, mode="mirror", truncate=4.0)
    return np.clip(np.floor((1.0 + amount) * f - amount * blur + 0.5), 0, 255).astype(np.uint8)
