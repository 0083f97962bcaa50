"""GPU tests of the drop-in surface (dct_svd_core_secure.embed/extract/detect)
and of the BASELINE configs at full size through size-independent properties."""
import glob
import os

import numpy as np
import pytest

from oracle import wm_oracle as o

pytestmark = pytest.mark.gpu

GOLDEN_T8 = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*_t8*.npz")))
GOLDEN_REF = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*_ref.npz")))


@pytest.fixture(scope="module")
def core(gpu_ctx):
    import dct_svd_core_secure as c
    return c


@pytest.mark.parametrize("path", GOLDEN_T8, ids=[os.path.basename(p)[:-4] for p in GOLDEN_T8])
def test_golden_fixture_through_the_dropin(core, path):
    g = np.load(path, allow_pickle=False)
    nonce = bytes(g["meta_nonce"].tolist())
    r = core.embed_arrays(g["cover"], g["wm"], "golden-pw", nonce, float(g["alpha"]), bool(g["color"]),
                          float(g["kfrac"]), 8, int(g["k_floor"]))
    d = np.abs(r["stego"].astype(int) - g["stego"].astype(int))
    # gray: a 1-LSB flip of Y can move B,G,R by up to 2 after YCrCb->BGR fixed point
    assert d.max() <= (1 if bool(g["color"]) else 2)
    assert np.mean(d != 0) < 5e-3
    assert abs(r["psnr"] - float(g["psnr"])) < 2e-2 and abs(r["ssim"] - float(g["ssim"])) < 1e-3
    for k in ("Sc", "Sw", "Sb", "Sg", "Sr", "SWb"):
        if "meta_" + k in g:
            a, b = r["meta"][k], g["meta_" + k]
            assert np.max(np.abs(a - b) / np.maximum(b[..., :1], 1e-30)) < 1e-4
    # interop both ways: GPU meta is readable by the oracle and vice versa (same keys, HMAC coverage)
    ex_o = o.extract_arrays(g["stego"], r["meta"], "golden-pw", True, 8, int(g["k_floor"]))
    gm = {k[5:]: g[k] for k in g.files if k.startswith("meta_")}
    gm["mode"] = "color" if bool(g["color"]) else "gray"
    gm["k_floor"] = g["k_floor"]          # the oracle takes k_floor as an argument, the product reads it from meta
    ex_g = core.extract_arrays(g["stego"], gm, "golden-pw", True)
    for ex in (ex_o, ex_g):
        assert np.mean(np.abs(ex.astype(int) - g["extracted"].astype(int)) > 1) < 2e-2
    ok, score = core.detect_arrays(g["stego"], gm, 0.6)
    assert ok and abs(score - float(g["detect_score"])) < 1e-3
    with pytest.raises(ValueError, match="Sai mật khẩu"):
        core.extract_arrays(g["stego"], gm, "wrong")


@pytest.mark.parametrize("color", [False, True])
def test_file_level_roundtrip(core, tmp_path, color):
    hg = __import__("importlib").import_module(core._impl.__package__ + ".hostglue")
    rng = np.random.default_rng(11)
    cover = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    wm = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    cp, wp = str(tmp_path / "cover.png"), str(tmp_path / "wm.png")
    assert hg.write_png(cp, cover) and hg.write_png(wp, wm)
    out, meta, ps, ss = core.embed(cp, wp, str(tmp_path / "result.jpg"), str(tmp_path / "m"),
                                   alpha=0.12, color=color, password="pw", nonce=bytes(8))
    assert out.endswith("result_stego.png") and os.path.exists(out)          # single:148-149
    assert os.path.exists(meta + ".npz")                                      # numpy appends .npz
    assert 15 < ps < 60 and 0.5 < ss <= 1.0
    ref = o.embed_arrays(cover, wm, "pw", bytes(8), 0.12, color, 0.6, 8)
    st = hg.read_image_bgr(out)
    assert np.abs(st.astype(int) - ref["stego"].astype(int)).max() <= 2
    assert abs(ps - ref["psnr"]) < 2e-2 and abs(ss - ref["ssim"]) < 1e-3
    data = np.load(meta + ".npz", allow_pickle=False)
    want = {"mode", "payload_type", "shape", "alpha", "kfrac", "nonce", "digest", "tile", "k_floor"}
    want |= {"Sb", "Sg", "Sr", "UWb", "VWbt", "SWb", "UWg", "VWgt", "SWg", "UWr", "VWrt", "SWr"} if color \
        else {"Sc", "Uw", "Vwt", "Sw"}
    assert set(data.files) == want
    wout = core.extract(out, meta + ".npz", str(tmp_path / "w.bmp"), password="pw", normalize=True)
    assert wout.endswith("w_wm.png") and os.path.exists(wout)                # single:225-226
    ok, score = core.detect(out, meta + ".npz")
    assert ok and score > 0.9
    ok2, s2 = core.detect(cp, meta + ".npz")                                  # the unmarked cover
    assert not ok2
    with pytest.raises(ValueError):
        core.extract(out, meta + ".npz", str(tmp_path / "x.png"), password="nope")
    with pytest.raises(ValueError):
        core.embed(cp, wp, out, meta, password="")
    # uncompressed meta: the same arrays, readable the same way
    _, meta_u, _, _ = core.embed(cp, wp, str(tmp_path / "result_u.png"), str(tmp_path / "mu"), alpha=0.12, color=color,
                                 password="pw", nonce=bytes(8), compress_meta=False)
    du = np.load(meta_u + ".npz", allow_pickle=False)
    assert set(du.files) == want and all(np.array_equal(du[k], data[k]) for k in want)
    assert core.detect(str(tmp_path / "result_u.png"), meta_u + ".npz")[0]
    with pytest.raises(ValueError):
        core.embed(str(tmp_path / "missing.png"), wp, out, meta, password="pw")
    with pytest.raises(ValueError):
        core.embed(cp, wp, out, meta, password="pw", tile=16)


def _frob_check(ctx, plane_u8, sigma):
    """Parseval per tile: sum_i sigma_i^2 == |tile|_F^2 (oracle-free, any size)."""
    H, W = plane_u8.shape
    t = plane_u8[: H // 8 * 8, : W // 8 * 8].astype(np.float64).reshape(H // 8, 8, W // 8, 8)
    fro = (t * t).sum(axis=(1, 3))
    s2 = (sigma.astype(np.float64) ** 2).sum(-1)
    assert np.max(np.abs(s2 - fro) / np.maximum(fro, 1.0)) < 1e-5
    assert np.all(np.diff(sigma, axis=-1) <= 1e-3 * sigma[..., :1])


@pytest.mark.parametrize("H,W,alpha", [(1080, 1920, 0.15), (2160, 3840, 0.15)])
def test_full_size_embed_extract_detect_properties(gpu_ctx, H, W, alpha):
    """BASELINE configs 2 (1080p Y) and the 4K metric shape, at full size."""
    host = np.random.default_rng(1234).integers(0, 256, (H, W), dtype=np.uint8)
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    stego, sc, _ = gpu_ctx.embed_tiles(host, S, alpha)
    _frob_check(gpu_ctx, host, sc)
    # alpha = 0: identity up to truncation; K = 0: same thing
    st0, sc0, _ = gpu_ctx.embed_tiles(host, S, 0.0)
    d0 = host.astype(int) - st0.astype(int)
    assert d0.min() >= -1 and d0.max() <= 1 and np.mean(d0 != 0) < 0.6
    stK0, _, _ = gpu_ctx.embed_tiles(host, S, alpha, K=0)
    assert np.array_equal(stK0, st0) and np.array_equal(sc0, sc)
    # oracle on a crop of full tiles (tiles are independent): exact parity there
    ch, cw = 64, 128
    ref = o.embed_plane(host[:ch, :cw].astype(np.float32), wys[:ch, :cw], alpha, 0.6, 8)
    assert np.abs(stego[:ch, :cw].astype(int) - ref["stego"].astype(int)).max() <= 1
    # round trip: extraction correlates with the scrambled watermark, detect says yes
    w = gpu_ctx.extract_tiles(stego, sc, U, Vt, alpha)
    assert np.corrcoef(w.ravel()[::7], wys.ravel()[::7])[0, 1] > 0.9
    assert gpu_ctx.detect_tiles(stego, sc, S, alpha)[0] > 0.95
    assert np.all(gpu_ctx.extract_tiles(stego, sc, U, Vt, alpha, K=0) == 0)
    # sigma-only kernel agrees with the embed kernel's side output on the same plane
    assert np.max(np.abs(gpu_ctx.sigma_tiles(host) - sc) / sc[..., :1]) < 1e-5


def test_config3_4k_colour_planes_shared_permutation(gpu_ctx):
    """BASELINE config 3: 3 planes (B,G,R) of 2160x3840, per-plane watermark sigma, alpha=0.18."""
    H, W, alpha = 2160, 3840, 0.18
    rng = np.random.default_rng(3)
    planes = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    wms = rng.integers(0, 256, (3, H, W)).astype(np.float32)
    U, S, Vt = gpu_ctx.svd_tiles(wms)
    stego, sc, _ = gpu_ctx.embed_tiles(planes, S, alpha)
    for p in range(3):
        _frob_check(gpu_ctx, planes[p], sc[p])
    for p in range(3):       # plane p must have used ITS watermark plane: oracle parity on a crop of each
        ref = o.embed_plane(planes[p, :32, :64].astype(np.float32), wms[p, :32, :64], alpha, 0.6, 8)
        assert np.abs(stego[p, :32, :64].astype(int) - ref["stego"].astype(int)).max() <= 1
        wrong = o.embed_plane(planes[p, :32, :64].astype(np.float32), wms[(p + 1) % 3, :32, :64], alpha, 0.6, 8)
        assert np.abs(stego[p, :32, :64].astype(int) - wrong["stego"].astype(int)).max() > 1
    scores = gpu_ctx.detect_tiles(stego, sc, S, alpha)
    assert scores.shape == (3,) and np.all(scores > 0.95)


def test_config5_8k_embed_extract_detect_kfloor_sweep(gpu_ctx):
    """BASELINE config 5: 4320x7680 single frame, mid-band sweep K = 1..8 (k_floor)."""
    H, W, alpha = 4320, 7680, 0.15
    host = np.random.default_rng(5).integers(0, 256, (H, W), dtype=np.uint8)
    wys = np.random.default_rng(6).integers(0, 256, (H, W)).astype(np.float32)
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    prev_psnr = 100.0
    for K in (1, 2, 4, 6, 8):
        stego, sc, _ = gpu_ctx.embed_tiles(host, S, alpha, K=K)
        ps = o.psnr(host, stego)
        assert ps <= prev_psnr + 1e-6            # perturbing more singular values can only cost PSNR
        prev_psnr = ps
        if K in (1, 8):
            ref = o.embed_plane(host[:16, :32].astype(np.float32), wys[:16, :32], alpha, 0.0, 8, k_floor=K)
            assert np.abs(stego[:16, :32].astype(int) - ref["stego"].astype(int)).max() <= 1
    _frob_check(gpu_ctx, host, sc)
    w = gpu_ctx.extract_tiles(stego, sc, U, Vt, alpha, K=8)
    assert np.corrcoef(w.ravel()[::31], wys.ravel()[::31])[0, 1] > 0.9
    assert gpu_ctx.detect_tiles(stego, sc, S, alpha)[0] > 0.95


@pytest.mark.parametrize("path", GOLDEN_REF, ids=[os.path.basename(p)[:-4] for p in GOLDEN_REF])
def test_fullframe_golden_fixture_through_the_dropin(core, path):
    """tile=None: the reference's own semantics.  Files written this way carry exactly
    the reference's meta keys, so the reference's extract/detect (here: the oracle's
    restatement of them) read them, and vice versa."""
    g = np.load(path, allow_pickle=False)
    nonce = bytes(g["meta_nonce"].tolist())
    color = bool(g["color"])
    r = core.embed_arrays(g["cover"], g["wm"], "golden-pw", nonce, float(g["alpha"]), color,
                          float(g["kfrac"]), None, int(g["k_floor"]))
    d = np.abs(r["stego"].astype(int) - g["stego"].astype(int))
    assert d.max() <= (1 if color else 2) and np.mean(d != 0) < 1e-2
    assert abs(r["psnr"] - float(g["psnr"])) < 5e-2 and abs(r["ssim"] - float(g["ssim"])) < 1e-3
    want = {"mode", "payload_type", "shape", "alpha", "kfrac", "nonce", "digest"}
    want |= {"Sb", "Sg", "Sr", "UWb", "VWbt", "SWb", "UWg", "VWgt", "SWg", "UWr", "VWrt", "SWr"} if color \
        else {"Sc", "Uw", "Vwt", "Sw"}
    assert set(r["meta"]) == want                                            # single:157-166,183-189
    for k in ("Sc", "Sw", "Sb", "SWb"):
        if "meta_" + k in g:
            a, b = r["meta"][k], g["meta_" + k]
            assert a.shape == b.shape and np.max(np.abs(a - b)) / b[0] < 1e-4
    H, W = g["cover"].shape[:2]
    L = min(H, W)
    U = r["meta"]["UWb" if color else "Uw"]; Vt = r["meta"]["VWbt" if color else "Vwt"]
    assert U.shape == (H, L) and Vt.shape == (L, W)
    # GPU-written files through the oracle's (= reference's) extract / detect
    ex_o = o.extract_arrays(r["stego"], r["meta"], "golden-pw", True, None, int(g["k_floor"]))
    ok_o, score_o = o.detect_arrays(r["stego"], r["meta"], 0.6, None)
    assert ok_o and abs(score_o - float(g["detect_score"])) < 5e-3
    assert np.mean(np.abs(ex_o.astype(int) - g["extracted"].astype(int)) > 2) < 5e-2
    # oracle-written (= reference-format) files through the GPU extract / detect
    if "meta_Uw" in g or "meta_UWb" in g:
        gm = {k[5:]: g[k] for k in g.files if k.startswith("meta_")}
        gm["mode"] = "color" if color else "gray"
        ex_g = core.extract_arrays(g["stego"], gm, "golden-pw", True)
        assert np.mean(np.abs(ex_g.astype(int) - g["extracted"].astype(int)) > 2) < 5e-2
        ok, score = core.detect_arrays(g["stego"], gm, 0.6)
        assert ok and abs(score - float(g["detect_score"])) < 5e-3
        with pytest.raises(ValueError, match="Sai mật khẩu"):
            core.extract_arrays(g["stego"], gm, "wrong")


def test_config4_256_frames_1080p_sharded_equals_unsharded(gpu_ctx, pkg):
    """BASELINE config 4 at full size: 256 frames 1080x1920, rank r of 8 takes frames
    [r*256//8, (r+1)*256//8).  Property: the ranks' shares tile the batch and are bit-identical
    to the unsharded run; two frames are checked against the oracle on a crop of full tiles."""
    import importlib
    v = importlib.import_module(pkg.__name__ + ".video")
    sh = importlib.import_module(pkg.__name__ + ".sharding")
    N, H, W, alpha = 256, 1080, 1920, 0.15
    rng = np.random.default_rng(1234)
    frames = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    Uw, Sw, Vwt = gpu_ctx.svd_tiles(wys)
    st_all, sc_all = v.embed_frames(gpu_ctx, frames, Sw, alpha, 8, batch=32)
    covered = []
    for r in range(8):
        (lo, hi), st, sc = v.embed_frames_sharded(gpu_ctx, frames, Sw, alpha, 8, rank=r, world_size=8, batch=32)
        assert (lo, hi) == sh.frame_range(r, 8, N) == (32 * r, 32 * r + 32)
        assert np.array_equal(st, st_all[lo:hi]) and np.array_equal(sc, sc_all[lo:hi])
        covered.extend(range(lo, hi))
    assert covered == list(range(N))
    for i in (0, 255):
        ref = o.embed_plane(frames[i, :64, :128].astype(np.float32), wys[:64, :128], alpha, 0.0, 8, k_floor=8)
        assert np.abs(st_all[i, :64, :128].astype(int) - ref["stego"].astype(int)).max() <= 1
    assert 20 < o.psnr(frames[7], st_all[7]) < 30


def test_dropin_is_reentrant_across_threads(core):
    """Two threads inside embed_arrays()/extract_arrays() at the same time (the reference's
    functions are plain re-entrant Python): each thread gets its own context, results equal
    the single-threaded ones."""
    import threading
    rng = np.random.default_rng(8)
    covers = [rng.integers(0, 256, (96, 128, 3), dtype=np.uint8) for _ in range(2)]
    wm = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    want = [core.embed_arrays(c, wm, "pw", bytes(8), alpha=0.12, tile=t) for c, t in zip(covers, (8, None))]
    got = [None, None]; errs = []

    def work(i, tile):
        try:
            for _ in range(4):
                got[i] = core.embed_arrays(covers[i], wm, "pw", bytes(8), alpha=0.12, tile=tile)
                w = core.extract_arrays(got[i]["stego"], got[i]["meta"], "pw")
                assert w.shape[:2] == (96, 128)
        except Exception as e:             # surfaced below; a thread must not die silently
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i, t)) for i, t in enumerate((8, None))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    for g, w in zip(got, want):
        assert np.array_equal(g["stego"], w["stego"])
        assert g["psnr"] == w["psnr"]


@pytest.mark.parametrize("tile", [8, None])
def test_array_level_roundtrip_on_odd_sizes(core, tile):
    """Odd, tiny, portrait and non-multiple-of-8 covers through embed/extract/detect_arrays in both
    modes and both colour settings: no exceptions, finite metrics, stego equal to the oracle's within
    1-2 LSB, the watermark detected where there is room for one (OpenCV's dct needs even sizes; the
    GPU path does not, so odd sizes are only checked for sanity, not against the oracle)."""
    rng = np.random.default_rng(21)
    wm = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    for (H, W) in ((64, 48), (37, 53), (16, 16), (9, 20), (130, 70), (8, 8)):
        for color in (False, True):
            cover = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            r = core.embed_arrays(cover, wm, "pw", bytes(8), alpha=0.15, color=color, tile=tile)
            st = r["stego"]
            assert st.shape == cover.shape and st.dtype == np.uint8
            assert np.isfinite(r["psnr"]) and np.isfinite(r["ssim"])
            if H % 2 == 0 and W % 2 == 0:
                ref = o.embed_arrays(cover, wm, "pw", bytes(8), 0.15, color, 0.6, tile)
                assert np.abs(st.astype(int) - ref["stego"].astype(int)).max() <= 2, (H, W, color, tile)
            w = core.extract_arrays(st, r["meta"], "pw")
            assert w.shape[:2] == (H, W)
            ok, score = core.detect_arrays(st, r["meta"])
            assert np.isfinite(score)
            if min(H, W) >= 16:
                assert ok and score > 0.8, (H, W, color, tile, score)


@pytest.mark.parametrize("tile", [8, None])
def test_config1_literal_512_gray_64_logo_alpha_012(core, tmp_path, tile):
    """BASELINE config 1 with its literal parameters - 512x512 grayscale host, 64x64 watermark (x8 replication by
    the INTER_AREA resize), alpha = 0.12, kfrac = 0.6 - through the file-level embed_watermark / extract_watermark /
    detect aliases, against the oracle's array pipeline on the same inputs, in both modes."""
    hg = __import__("importlib").import_module(core._impl.__package__ + ".hostglue")
    g = np.random.default_rng(1234).integers(0, 256, (512, 512), dtype=np.uint8)
    cover = np.stack([g] * 3, axis=-1)                                  # a grayscale host as cv2.imread(IMREAD_COLOR) returns it
    wm1 = np.random.default_rng(4321).integers(0, 256, (64, 64), dtype=np.uint8)
    wm = np.stack([wm1] * 3, axis=-1)
    cp, wp = str(tmp_path / "host.png"), str(tmp_path / "logo.png")
    assert hg.write_png(cp, cover) and hg.write_png(wp, wm)
    out, meta, ps, ss = core.embed_watermark(cp, wp, str(tmp_path / "stego.png"), str(tmp_path / "meta.npz"), alpha=0.12,
                                             color=False, password="bench", nonce=bytes(8), tile=tile)
    ref = o.embed_arrays(cover, wm, "bench", bytes(8), 0.12, False, 0.6, tile)
    st = hg.read_image_bgr(out)
    d = np.abs(st.astype(int) - ref["stego"].astype(int))
    assert d.max() <= 2 and np.mean(d != 0) < 5e-3                       # 1 LSB on Y, up to 2 after YCrCb -> BGR
    assert abs(ps - ref["psnr"]) < 2e-2 and abs(ss - ref["ssim"]) < 1e-3
    data = np.load(meta, allow_pickle=False)
    sc, sco = data["Sc"], ref["meta"]["Sc"]
    assert np.max(np.abs(sc - sco) / np.maximum(sco[..., :1] if sco.ndim > 1 else sco[0], 1e-30)) < 1e-4
    assert np.array_equal(hg.resize_area(wm, 512, 512), np.kron(wm1, np.ones((8, 8), np.uint8))[..., None].repeat(3, -1))
    wout = core.extract_watermark(out, meta, str(tmp_path / "wm_out.png"), password="bench")
    ex_o = o.extract_arrays(ref["stego"], ref["meta"], "bench", True, tile)
    ex = hg.read_image_bgr(wout)[..., 0]
    assert np.mean(np.abs(ex.astype(int) - ex_o.astype(int)) > 2) < 2e-2
    ok, score = core.detect(out, meta)
    so = o.detect_arrays(ref["stego"], ref["meta"], 0.6, tile)[1]
    assert ok and abs(score - so) < 2e-3


@pytest.mark.parametrize("tile", [8, None])
def test_array_level_roundtrip_on_generated_scenes(core, tile):
    """The drop-in on UI-like covers (flat areas, rectangles, rules, gradients - rank-deficient tiles / planes of every
    kind) and a structured logo, gray and colour: no exception (in particular no 'SVD did not converge' from a
    rank-deficient tile), finite metrics, detection equal to the oracle's score where the oracle's own arithmetic defines
    it, the watermark recovered like the oracle recovers it."""
    from test_gpu_parity import _scene
    rng = np.random.default_rng(4242)
    logo = np.zeros((24, 40, 3), np.uint8); logo[4:20, 6:34] = (30, 200, 120); logo[8:16, 12:28] = 255; logo[::5] = 0
    for case in range(5):
        H = int(rng.choice([64, 96, 128])); W = int(rng.choice([96, 128, 192]))
        cover = np.stack([_scene(rng, H, W) for _ in range(3)], -1)
        for color in (False, True):
            r = core.embed_arrays(cover, logo, "pw", bytes(range(8)), alpha=0.12, color=color, tile=tile)
            st = r["stego"]
            assert st.shape == cover.shape and np.isfinite(r["psnr"]) and np.isfinite(r["ssim"]), (case, color)
            w = core.extract_arrays(st, r["meta"], "pw")
            ok, score = core.detect_arrays(st, r["meta"])
            assert np.isfinite(score), (case, color)
            ref = o.embed_arrays(cover, logo, "pw", bytes(range(8)), 0.12, color, 0.6, tile)
            ok_o, score_o = o.detect_arrays(ref["stego"], ref["meta"], tile=tile)
            assert abs(r["psnr"] - ref["psnr"]) < 0.5, (case, color, r["psnr"], ref["psnr"])
            assert abs(score - score_o) < 5e-2, (case, color, tile, score, score_o)
            w_o = o.extract_arrays(ref["stego"], ref["meta"], "pw", tile=tile)
            a = w.astype(np.float64).ravel(); b = w_o.astype(np.float64).ravel()
            if a.std() > 0 and b.std() > 0:
                assert np.corrcoef(a, b)[0, 1] > 0.5, (case, color, tile, np.corrcoef(a, b)[0, 1])


@pytest.mark.parametrize("color", [False, True], ids=["gray", "colour"])
def test_fullframe_resized_stego_follows_the_reference(core, color):
    """A stego that is not the size its meta was written for (cropped, padded or rescaled - the usual robustness
    experiment).  The reference does not look at the size: extract takes sigma of whatever plane it is given, cuts to
    L = min(len(Sc), len(S_cw), Uw.shape[0], Vwt.shape[0]), uses the [:L,:L] corner of the meta's factors and the
    META's H x W for the zero plane and the permutation (single:205-220); detect cuts the vectors to the shortest
    (single:299).  The drop-in must do the same (it raised ValueError until round 3b); oracle = restatement of those lines."""
    rng = np.random.default_rng(77)
    H, W = 96, 128
    yy, xx = np.mgrid[0:H, 0:W]
    base = 110 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    cover = np.clip(base[..., None] + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    wm = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    nonce = bytes(range(8))
    r = core.embed_arrays(cover, wm, "pw", nonce, 0.15, color, 0.6, None, 8)
    stego, meta = r["stego"], r["meta"]
    same = core.extract_arrays(stego, meta, "pw", True)
    variants = {
        "cropped": np.ascontiguousarray(stego[8:88, 10:122]),                                   # 80 x 112: L = 80 < 96
        "padded": np.pad(stego, ((0, 16), (0, 32), (0, 0)), mode="edge"),                       # 112 x 160: L stays 96
        "tall": np.ascontiguousarray(np.repeat(stego, 2, axis=0)[:160, :100]),                  # 160 x 100: transposed aspect
    }
    for name, st in variants.items():
        ex_o = o.extract_arrays(st, meta, "pw", True, None, 8)
        ex_g = core.extract_arrays(st, meta, "pw", True)
        assert ex_g.shape == ex_o.shape == same.shape, name          # the META's size, whatever the stego's
        d = np.abs(ex_g.astype(int) - ex_o.astype(int))
        assert np.mean(d > 2) < 2e-2, (name, float(np.mean(d > 2)), int(d.max()))
        ok_o, sc_o = o.detect_arrays(st, meta, 0.6, None)
        ok_g, sc_g = core.detect_arrays(st, meta, 0.6)
        assert abs(sc_g - sc_o) < 2e-3 and ok_g == ok_o, (name, sc_g, sc_o)
    # the same-size path is untouched by the new branch
    ex_o = o.extract_arrays(stego, meta, "pw", True, None, 8)
    assert np.mean(np.abs(same.astype(int) - ex_o.astype(int)) > 2) < 2e-2
    # tile mode still names the mismatch (its meta holds per-tile factors of one size)
    r8 = core.embed_arrays(cover, wm, "pw", nonce, 0.15, color, 0.6, 8, 8)
    with pytest.raises(ValueError, match="meta was written for"):
        core.extract_arrays(variants["cropped"], r8["meta"], "pw", True)
    with pytest.raises(ValueError, match="meta was written for"):
        core.detect_arrays(variants["cropped"], r8["meta"], 0.6)


def test_ref_reconstruct_against_numpy(gpu_ctx):
    """wm_ref_reconstruct_f32 = single:214-218 with the estimates given, for L below, at and above nothing: the [:L,:L]
    corner of the factors, zero elsewhere, idct2 of the whole plane."""
    rng = np.random.default_rng(5)
    for (H, W) in ((64, 96), (96, 64), (48, 48)):
        Lm = min(H, W)
        Uw = rng.normal(0, 1, (H, Lm)).astype(np.float32); Vwt = rng.normal(0, 1, (Lm, W)).astype(np.float32)
        for L in (0, 1, Lm // 2 + 1, Lm):
            sh = rng.normal(0, 30, L).astype(np.float32)
            full = np.zeros((H, W), np.float64)
            full[:L, :L] = (Uw[:L, :L].astype(np.float64) * sh.astype(np.float64)) @ Vwt[:L, :L].astype(np.float64)
            want = o.idct2(full.astype(np.float32))
            got = gpu_ctx.ref_reconstruct(Uw, sh, Vwt, H, W)
            assert got.shape == (H, W)
            assert np.max(np.abs(got - want)) <= 2e-4 * max(1.0, float(np.abs(want).max())), (H, W, L)
    with pytest.raises(ValueError):
        gpu_ctx.ref_reconstruct(Uw, np.zeros(Lm + 1, np.float32), Vwt, H, W)
