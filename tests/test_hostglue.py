"""Host glue of the drop-in (colour, resize, permutation, HMAC, metrics, PNG
I/O) against the oracle's independent restatement and NumPy/hashlib known answers."""
import importlib
import os

import numpy as np
import pytest

from conftest import PKG_NAME
from oracle import wm_oracle as o

hg = importlib.import_module(PKG_NAME + ".hostglue")


def test_colour_and_resize_match_oracle():
    a = np.random.default_rng(0).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    assert np.array_equal(hg.bgr_to_gray(a), o.bgr_to_gray(a))
    assert np.array_equal(hg.bgr_to_ycrcb(a), o.bgr_to_ycrcb(a))
    assert np.array_equal(hg.ycrcb_to_bgr(a), o.ycrcb_to_bgr(a))
    for (W, H) in ((112, 80), (28, 20), (50, 33), (56, 40), (28, 80), (70, 25)):   # last two: one axis shrinks, one enlarges
        assert np.array_equal(hg.resize_area(a, W, H), o.resize_area(a, W, H)), (W, H)
    # known answers: saturation + exact grey
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 0, 255]]], np.uint8)
    assert hg.bgr_to_gray(px).tolist() == [[255, 0, 29, 76]]
    assert hg.bgr_to_ycrcb(px)[0, 0].tolist() == [255, 128, 128]


def test_security_matches_oracle_and_numpy():
    key = hg.derive_key("bench", bytes(8))
    assert key == o.derive_key("bench", bytes(8))
    idx = hg.permutation_index(24, 40, key)
    assert np.array_equal(idx, o.permutation(24, 40, o.rng_from_key(key)))
    x = np.random.default_rng(1).uniform(0, 255, (24, 40)).astype(np.float32)
    assert np.array_equal(hg.unpermute(hg.permute(x, idx), idx), x)
    parts = [np.arange(5, dtype=np.float32), np.ones((2, 3), np.float32)]
    assert hg.hmac_digest(key, parts) == o.hmac_digest(key, [p.tobytes() for p in parts])
    assert hg.digests_equal(b"ab", b"ab") and not hg.digests_equal(b"ab", b"ac")


def test_metrics_match_oracle():
    a = np.random.default_rng(2).integers(0, 256, (48, 64, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + np.random.default_rng(3).integers(-6, 7, a.shape), 0, 255).astype(np.uint8)
    assert abs(hg.psnr(a, b) - o.psnr(a, b)) < 1e-9
    assert abs(hg.ssim(a, b) - o.ssim(a, b)) < 1e-6
    assert hg.psnr(a, a) == 99.0
    x = np.random.default_rng(4).normal(0, 50, (8, 8)).astype(np.float32)
    assert np.allclose(hg.normalize_minmax(x), o.normalize_minmax(x))


def test_png_roundtrip_and_errors(tmp_path):
    a = np.random.default_rng(5).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    assert hg.write_png(p, a, 0)
    assert np.array_equal(hg.read_image_bgr(p), a)
    g = a[..., 0].copy()
    assert hg.write_png(str(tmp_path / "g.png"), g, 1)
    assert np.array_equal(hg.read_image_bgr(str(tmp_path / "g.png")), np.stack([g] * 3, -1))   # IMREAD_COLOR
    with pytest.raises(ValueError):
        hg.read_image_bgr(str(tmp_path / "missing.png"))
    assert not hg.write_png(str(tmp_path / "no_such_dir" / "x.png"), a, 0)


def test_permutation_index_cache_and_resize_speed():
    """The keyed shuffle is cached per (H, W, key) (read-only array, same object on a hit, evicted
    after two other keys); the watermark resize of a small colour logo onto a large cover must stay
    a BLAS-sized job (it once was a 70 s einsum)."""
    import time
    k1, k2, k3 = (o.derive_key(p, bytes(8)) for p in ("a", "b", "c"))
    i1 = hg.permutation_index(40, 56, k1)
    assert hg.permutation_index(40, 56, k1) is i1 and not i1.flags.writeable
    assert np.array_equal(i1, o.permutation(40, 56, o.rng_from_key(k1)))
    hg.permutation_index(40, 56, k2); hg.permutation_index(40, 56, k3)
    i1b = hg.permutation_index(40, 56, k1)
    assert i1b is not i1 and np.array_equal(i1b, i1)                # evicted, recomputed identically
    assert np.array_equal(hg.permutation_index(56, 40, k1), i1)    # the shuffle only knows H*W
    wm = np.random.default_rng(0).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    t0 = time.perf_counter()
    r = hg.resize_area(wm, 1000, 600)
    assert time.perf_counter() - t0 < 5.0 and r.shape == (600, 1000, 3)
    assert np.array_equal(r, o.resize_area(wm, 1000, 600))


def test_read_image_modes(tmp_path):
    """cv2.imread(IMREAD_COLOR) semantics for what Pillow hands back: gray, gray+alpha, RGBA, palette,
    1-bit and 16-bit PNGs all come out as 3-channel 8-bit BGR."""
    from PIL import Image
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)
    g = rgb[..., 0]
    cases = {
        "L": (Image.fromarray(g), np.repeat(g[..., None], 3, 2)),
        "RGB": (Image.fromarray(rgb), rgb[..., ::-1]),
        "RGBA": (Image.fromarray(np.dstack([rgb, g])), rgb[..., ::-1]),
        "LA": (Image.fromarray(np.dstack([g, rgb[..., 1]]), "LA"), np.repeat(g[..., None], 3, 2)),
        "I16": (Image.fromarray(g.astype(np.uint16) * 257), np.repeat(g[..., None], 3, 2)),
        "1": (Image.fromarray(g > 128), np.repeat(((g > 128) * 255).astype(np.uint8)[..., None], 3, 2)),
    }
    for name, (im, want) in cases.items():
        p = str(tmp_path / (name + ".png")); im.save(p)
        got = hg.read_image_bgr(p)
        assert got.dtype == np.uint8 and got.shape == (12, 18, 3) and got.flags.c_contiguous, name
        assert np.array_equal(got, want), name


def test_resize_cache_returns_the_same_pixels_and_tracks_content():
    """resize_area_cached is resize_area memoised on the image's bytes: equal pixels, a changed logo is a miss,
    and the shared result cannot be written through."""
    rng = np.random.default_rng(5)
    wm = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    a = hg.resize_area_cached(wm, 100, 60)
    assert np.array_equal(a, hg.resize_area(wm, 100, 60))
    assert hg.resize_area_cached(wm.copy(), 100, 60) is a            # same content: a hit
    wm2 = wm.copy(); wm2[3, 4, 1] ^= 1
    b = hg.resize_area_cached(wm2, 100, 60)
    assert b is not a and np.array_equal(b, hg.resize_area(wm2, 100, 60))
    assert not a.flags.writeable
    assert np.array_equal(hg.resize_area_cached(wm, 60, 100), hg.resize_area(wm, 60, 100))   # other size: its own entry


def test_non_string_password_is_a_type_error_not_a_key():
    """The legacy module of the same name had extract(stego, meta, out, normalize=True) without a password
    (dct_svd_core_secure.py:203 of the reference): a positional True from such a call site must not be hashed
    as a password.  Checked before anything touches a file or the GPU."""
    import importlib
    import pytest
    m = importlib.import_module("dct_svd_core_secure")
    with pytest.raises(TypeError, match="password must be a str"):
        m.extract("stego.png", "meta.npz", "out.png", True)
    with pytest.raises(TypeError, match="password must be a str"):
        m.embed("cover.png", "wm.png", "out.png", "meta.npz", 0.05, False, 7)
    with pytest.raises(TypeError):
        m.embed_arrays(np.zeros((8, 8, 3), np.uint8), np.zeros((8, 8, 3), np.uint8), b"bytes", bytes(8))
    with pytest.raises(ValueError):                      # the reference's own check is unchanged (single:115-116)
        m.embed("cover.png", "wm.png", "out.png", "meta.npz", password="")
    with pytest.raises(ValueError):
        m.extract("stego.png", "meta.npz", "out.png", None)


def test_save_npz_is_an_ordinary_compressed_npz(tmp_path):
    """hostglue.save_npz writes what np.savez_compressed writes (single:157-166,183-189) - same member names, dtypes, shapes
    and values for np.load, the reference's reader -, with the members deflated on worker threads; '.npz' is appended like
    NumPy does; strings, 0-d and empty arrays included; uncompressed and oversized inputs go through NumPy itself."""
    import zipfile
    from conftest import PKG_NAME
    hg = __import__("importlib").import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(4)
    meta = dict(mode="gray", payload_type="image", Sc=rng.normal(0, 1, (9, 7, 8)).astype(np.float32),
                Uw=rng.normal(0, 1, (9, 7, 8, 8)).astype(np.float32), Vwt=rng.normal(0, 1, (9, 7, 8, 8)).astype(np.float32)[:, ::-1],
                shape=np.array((72, 56)), alpha=0.12, kfrac=0.6, nonce=np.frombuffer(bytes(range(8)), dtype=np.uint8),
                tile=np.int32(8), empty=np.zeros((0, 3), np.float32), big=rng.integers(0, 9, 300000).astype(np.int64))
    p = hg.save_npz(str(tmp_path / "meta"), meta)
    assert p.endswith("meta.npz") and zipfile.ZipFile(p).testzip() is None
    np.savez_compressed(str(tmp_path / "ref.npz"), **meta)
    a, b = np.load(p, allow_pickle=False), np.load(str(tmp_path / "ref.npz"), allow_pickle=False)
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert all(i.compress_type == zipfile.ZIP_DEFLATED for i in zipfile.ZipFile(p).infolist())
    q = hg.save_npz(str(tmp_path / "plain.npz"), meta, compressed=False)
    c = np.load(q, allow_pickle=False)
    assert all(np.array_equal(c[k], b[k]) for k in b.files)
    assert all(i.compress_type == zipfile.ZIP_STORED for i in zipfile.ZipFile(q).infolist())


def test_load_npz_reads_every_member(tmp_path):
    from conftest import PKG_NAME
    hg = __import__("importlib").import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(6)
    meta = dict(mode="color", Sb=rng.normal(0, 1, 40).astype(np.float32), UWb=rng.normal(0, 1, (64, 40)).astype(np.float32),
                VWbt=rng.normal(0, 1, (40, 96)).astype(np.float32), alpha=0.2, nonce=np.frombuffer(bytes(8), dtype=np.uint8))
    for comp in (True, False):
        p = hg.save_npz(str(tmp_path / f"m{int(comp)}"), meta, compressed=comp)
        d = hg.load_npz(p)
        assert sorted(d) == sorted(meta) and str(d["mode"]) == "color" and float(d["alpha"]) == 0.2
        assert all(np.array_equal(d[k], np.asarray(meta[k])) for k in meta)
    with pytest.raises(FileNotFoundError):
        hg.load_npz(str(tmp_path / "missing.npz"))


def test_write_png_round_trips_through_an_independent_decoder(tmp_path):
    """hostglue.write_png builds the PNG itself (filter type 0, one zlib stream, level 0 = stored blocks like the reference's
    stego files): Pillow - an independent decoder - must read back the same pixels for gray and BGR, odd sizes and every level."""
    from PIL import Image
    from conftest import PKG_NAME
    hg = __import__("importlib").import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(8)
    for shape in ((1, 1), (7, 13), (64, 96), (33, 1), (5, 9, 3), (120, 67, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        for level in (0, 1, 6):
            p = str(tmp_path / f"a_{len(shape)}_{shape[0]}_{level}.png")
            assert hg.write_png(p, img, level)
            with Image.open(p) as im:
                im.load()
                got = np.asarray(im)
                assert im.mode == ("L" if img.ndim == 2 else "RGB")
            want = img if img.ndim == 2 else img[..., ::-1]
            assert got.shape == want.shape and np.array_equal(got, want)
            back = hg.read_image_bgr(p)
            assert np.array_equal(back, img if img.ndim == 3 else np.repeat(img[..., None], 3, axis=2))
    assert hg.write_png(str(tmp_path / "no_such_dir" / "x.png"), img, 0) is False


def test_read_image_fast_path_equals_pillow(tmp_path):
    """read_image_bgr takes PNGs whose scanlines all carry filter type 0 (what write_png produces) without Pillow; every
    other PNG - filtered scanlines, palette, 16-bit, interlaced, RGBA - must come out exactly as Pillow decodes it."""
    from PIL import Image
    from conftest import PKG_NAME
    hg = __import__("importlib").import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:70, 0:90]
    smooth = np.clip(100 + 60 * np.sin(xx / 9.0) + yy, 0, 255).astype(np.uint8)
    bgr = np.stack([smooth, smooth[::-1], rng.integers(0, 256, smooth.shape, dtype=np.uint8)], axis=-1)
    own = str(tmp_path / "own.png"); assert hg.write_png(own, bgr, 0)
    assert hg._read_png_unfiltered(own) is not None and np.array_equal(hg.read_image_bgr(own), bgr)
    own_g = str(tmp_path / "own_g.png"); assert hg.write_png(own_g, smooth, 1)
    assert np.array_equal(hg.read_image_bgr(own_g), np.repeat(smooth[..., None], 3, axis=2))
    pil = str(tmp_path / "pil.png"); Image.fromarray(bgr[..., ::-1].copy()).save(pil, compress_level=6)      # adaptive filters
    assert np.array_equal(hg.read_image_bgr(pil), bgr)
    pal = str(tmp_path / "pal.png"); Image.fromarray(smooth).convert("P").save(pal)
    want = np.asarray(Image.open(pal).convert("RGB"))[..., ::-1]
    assert hg._read_png_unfiltered(pal) is None and np.array_equal(hg.read_image_bgr(pal), want)
    rgba = str(tmp_path / "rgba.png"); Image.fromarray(np.dstack([bgr[..., ::-1], smooth])).save(rgba)
    assert hg._read_png_unfiltered(rgba) is None and np.array_equal(hg.read_image_bgr(rgba), bgr)
    inter = str(tmp_path / "bad.png"); open(inter, "wb").write(open(own, "rb").read()[:200])                  # truncated file
    with pytest.raises(ValueError, match="Không mở được ảnh"):
        hg.read_image_bgr(inter)


def test_save_npz_members_in_any_memory_order_and_pathlike(tmp_path):
    """a Fortran-ordered or transposed member must load back as the same array (the header has to describe the
    C-ordered bytes that are written), 0-d and empty members keep their shape, and os.PathLike targets work like np.savez's."""
    import pathlib
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(3)
    a = rng.normal(size=(37, 53)).astype(np.float32)
    arrays = {"c": a, "f": np.asfortranarray(a), "t": a.T, "strided": a[::2, 1::3], "f3": np.asfortranarray(rng.normal(size=(4, 5, 6))),
              "scalar": np.float32(0.12), "empty": np.zeros((0, 8), np.float32), "i": np.arange(12, dtype=np.int64).reshape(3, 4).T}
    out = hg.save_npz(pathlib.Path(tmp_path) / "meta_any_order", arrays)
    assert out.endswith("meta_any_order.npz") and os.path.exists(out)
    with np.load(out, allow_pickle=False) as z:
        assert sorted(z.files) == sorted(arrays)
        for k, v in arrays.items():
            got = z[k]
            assert got.shape == np.asarray(v).shape and got.dtype == np.asarray(v).dtype, k
            assert np.array_equal(got, np.asarray(v)), k


def _png_chunks(data):
    import struct
    pos, out = 8, []
    while pos + 12 <= len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        out.append((data[pos + 4:pos + 8], pos, n))
        pos += 12 + n
    return out


def test_png_fast_path_refuses_damaged_and_oversized_streams(tmp_path):
    """the module's own PNG reader verifies chunk CRCs, inflates with a bound of the header's size and honours
    Pillow's pixel limit; what it refuses falls through to Pillow, whose error becomes the reference's ValueError."""
    import struct
    import zlib
    from PIL import Image
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    img = np.random.default_rng(1).integers(0, 256, (24, 40, 3), dtype=np.uint8)
    good = str(tmp_path / "good.png")
    assert hg.write_png(good, img)
    assert np.array_equal(hg._read_png_unfiltered(good), img)
    data = bytearray(open(good, "rb").read())
    # 1. one flipped byte inside IDAT: the chunk CRC no longer matches -> not the fast path's business
    tag, pos, n = [c for c in _png_chunks(bytes(data)) if c[0] == b"IDAT"][0]
    bad = bytearray(data); bad[pos + 8 + n // 2] ^= 0x40
    p_bad = str(tmp_path / "badcrc.png"); open(p_bad, "wb").write(bytes(bad))
    assert hg._read_png_unfiltered(p_bad) is None
    with pytest.raises(ValueError):
        hg.read_image_bgr(p_bad)                      # Pillow refuses it as well: the reference's "cannot open" error
    # 2. an IDAT that inflates to far more than the header announces (a 24 x 40 header over 8 MB of zeros, valid CRCs)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(d, zlib.crc32(t)) & 0xFFFFFFFF)
    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 40, 24, 8, 2, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(bytes(8 << 20), 9)) + chunk(b"IEND", b"")
    p_bomb = str(tmp_path / "bomb.png"); open(p_bomb, "wb").write(bomb)
    import tracemalloc
    tracemalloc.start()
    assert hg._read_png_unfiltered(p_bomb) is None
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < (1 << 20)                           # never materialised the 8 MB
    # 3. a header over Pillow's pixel limit is left to Pillow's own guard
    old = Image.MAX_IMAGE_PIXELS
    try:
        Image.MAX_IMAGE_PIXELS = 100
        assert hg._read_png_unfiltered(good) is None
    finally:
        Image.MAX_IMAGE_PIXELS = old
    # 4. a truncated file
    p_tr = str(tmp_path / "trunc.png"); open(p_tr, "wb").write(bytes(data[:len(data) - 20]))
    assert hg._read_png_unfiltered(p_tr) is None


def test_nc_of_empty_vectors_is_zero():
    """single:286: an empty Sw / S in a meta gives 0.0 (detect -> False), not NaN."""
    import ast
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), PKG_NAME, "dct_svd_core_secure.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "_nc"][0]
    ns = {"np": np}
    exec(compile(ast.Module([fn], []), "_nc", "exec"), ns)
    assert ns["_nc"](np.zeros(0, np.float32), np.zeros(0, np.float32)) == 0.0
    assert ns["_nc"](np.zeros(0), np.arange(4.0)) == 0.0
    assert abs(ns["_nc"](np.arange(8.0), 2 * np.arange(8.0) + 1) - 1.0) < 1e-6
