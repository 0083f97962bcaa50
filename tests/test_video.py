"""Video frame loop: Y4M container (CPU) and the batched embed / averaged
extract / detect over frames (GPU)."""
import importlib
import os

import numpy as np
import pytest

from conftest import PKG_NAME
from oracle import wm_oracle as o


def _video(tmp_path, n=9, H=64, W=96, seed=3):
    rng = np.random.default_rng(seed)
    ys = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    chroma = rng.integers(0, 256, (n, 2 * (H // 2) * (W // 2)), dtype=np.uint8)
    v = importlib.import_module(PKG_NAME + ".video")
    p = str(tmp_path / "in.y4m")
    v.write_y4m(p, ys, chroma)
    return v, p, ys, chroma


def test_y4m_roundtrip(tmp_path):
    v, p, ys, chroma = _video(tmp_path)
    vid = v.Y4M(p)
    assert (vid.W, vid.H, vid.chroma) == (96, 64, "420")
    got = [(y.copy(), c.copy()) for _, y, c in vid]
    vid.close()
    assert len(got) == 9
    for i, (y, c) in enumerate(got):
        assert np.array_equal(y, ys[i]) and np.array_equal(c, chroma[i])
    mono = str(tmp_path / "m.y4m")
    v.write_y4m(mono, ys[:2])
    vm = v.Y4M(mono); frames = list(vm); vm.close()
    assert len(frames) == 2 and frames[0][2].size == 0
    open(str(tmp_path / "bad.y4m"), "wb").write(b"RIFFxxxx")
    with pytest.raises(ValueError):
        v.Y4M(str(tmp_path / "bad.y4m"))
    sh = importlib.import_module(PKG_NAME + ".sharding")
    assert [sh.frame_range(r, 4, 9) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 9)]
    # 4:4:4 (the colour functions' container): full-resolution chroma planes, Cb then Cr
    c444 = np.random.default_rng(2).integers(0, 256, (2, 2 * 64 * 96), dtype=np.uint8)
    p444 = str(tmp_path / "c444.y4m")
    v.write_y4m(p444, ys[:2], c444, chroma_tag="444")
    v4 = v.Y4M(p444); fr = [(y.copy(), c.copy()) for _, y, c in v4]; v4.close()
    assert v4.chroma == "444" and (v4.cw, v4.ch) == (96, 64) and len(fr) == 2
    assert np.array_equal(fr[1][0], ys[1]) and np.array_equal(fr[1][1], c444[1])


@pytest.mark.gpu
def test_video_embed_extract_detect(tmp_path, gpu_ctx):
    v, p, ys, chroma = _video(tmp_path)
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    wm = np.random.default_rng(5).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    wp = str(tmp_path / "wm.png"); assert hg.write_png(wp, wm)
    outp, meta, ps = v.embed_watermark_video(p, wp, str(tmp_path / "out.y4m"), str(tmp_path / "vm.npz"),
                                             alpha=0.15, frame_interval=2, password="pw", nonce=bytes(8), batch=2)
    assert 15 < ps < 60
    data = np.load(meta, allow_pickle=False)
    assert data["Sc"].shape == (5, 8, 12, 8) and int(data["frame_interval"]) == 2 and int(data["n_frames"]) == 9
    vid = v.Y4M(outp); got = [(y.copy(), c.copy()) for _, y, c in vid]; vid.close()
    assert len(got) == 9
    # oracle: same watermark preparation, per-frame tile-mode embed with the shared decomposition
    key = o.derive_key("pw", bytes(8)); idx = o.permutation(64, 96, o.rng_from_key(key))
    wy_s = o.permute(o.bgr_to_gray(o.resize_area(wm, 96, 64)).astype(np.float32), idx)
    wm_svd = o.watermark_decompose(wy_s, 8)
    for i, (y, c) in enumerate(got):
        assert np.array_equal(c, chroma[i])                               # chroma untouched
        if i % 2:
            assert np.array_equal(y, ys[i])                               # unmarked frames untouched
        else:
            ref = o.embed_plane(ys[i].astype(np.float32), wy_s, 0.15, 0.6, 8, wm_svd=wm_svd)
            assert np.abs(y.astype(int) - ref["stego"].astype(int)).max() <= 1
            assert np.max(np.abs(data["Sc"][i // 2] - ref["Sc"]) / ref["Sc"][..., :1]) < 1e-4
    ok, mean, scores = v.detect_watermark_video(outp, meta)
    assert ok and mean > 0.9 and scores.shape == (5,)
    ok0, mean0, _ = v.detect_watermark_video(p, meta)                      # the unmarked video
    assert not ok0
    wout = v.extract_watermark_video(outp, meta, str(tmp_path / "w.png"), password="pw")
    ex = hg.read_image_bgr(wout)[..., 0]
    want = o.bgr_to_gray(o.resize_area(wm, 96, 64))
    assert np.corrcoef(ex.ravel().astype(float), want.ravel().astype(float))[0, 1] > 0.8
    with pytest.raises(ValueError, match="Sai mật khẩu"):
        v.extract_watermark_video(outp, meta, str(tmp_path / "x.png"), password="nope")
    with pytest.raises(ValueError):
        v.embed_watermark_video(p, wp, outp, meta, password="")
    # array-level sharded embed: two ranks' shares tile the batch and equal the unsharded result
    st_all, sc_all = v.embed_frames(gpu_ctx, ys, wm_svd[1], 0.15)
    parts = [v.embed_frames_sharded(gpu_ctx, ys, wm_svd[1], 0.15, rank=r, world_size=2) for r in range(2)]
    assert parts[0][0] == (0, 4) and parts[1][0] == (4, 9)
    assert np.array_equal(np.concatenate([parts[0][1], parts[1][1]]), st_all)
    assert np.array_equal(np.concatenate([parts[0][2], parts[1][2]]), sc_all)


@pytest.mark.gpu
def test_video_fullframe_mode(tmp_path, gpu_ctx):
    """tile=None: every marked frame gets the reference's full-plane embed (one SVD per frame,
    batched); the meta carries per-frame Sc [n, L] and the full-plane Uw/Vwt."""
    v, p, ys, chroma = _video(tmp_path)
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    wm = np.random.default_rng(5).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    wp = str(tmp_path / "wm.png"); assert hg.write_png(wp, wm)
    outp, meta, ps = v.embed_watermark_video(p, wp, str(tmp_path / "out_ff.y4m"), str(tmp_path / "vm_ff.npz"),
                                             alpha=0.15, frame_interval=2, password="pw", nonce=bytes(8), batch=3,
                                             tile=None)
    data = np.load(meta, allow_pickle=False)
    assert int(data["tile"]) == 0 and data["Sc"].shape == (5, 64) and data["Uw"].shape == (64, 64)
    assert data["Vwt"].shape == (64, 96)
    vid = v.Y4M(outp); got = [(y.copy(), c.copy()) for _, y, c in vid]; vid.close()
    key = o.derive_key("pw", bytes(8)); idx = o.permutation(64, 96, o.rng_from_key(key))
    wy_s = o.permute(o.bgr_to_gray(o.resize_area(wm, 96, 64)).astype(np.float32), idx)
    for i, (y, c) in enumerate(got):
        assert np.array_equal(c, chroma[i])
        if i % 2:
            assert np.array_equal(y, ys[i])
        else:
            ref = o.embed_plane(ys[i].astype(np.float32), wy_s, 0.15, 0.6, None)
            assert np.abs(y.astype(int) - ref["stego"].astype(int)).max() <= 1
            assert np.mean(y != ref["stego"]) < 5e-3
            assert np.max(np.abs(data["Sc"][i // 2] - ref["Sc"])) / ref["Sc"][0] < 1e-5
    ok, mean, scores = v.detect_watermark_video(outp, meta)
    assert ok and mean > 0.9 and scores.shape == (5,)
    ok0, _, _ = v.detect_watermark_video(p, meta)
    assert not ok0
    wout = v.extract_watermark_video(outp, meta, str(tmp_path / "w_ff.png"), password="pw")
    ex = hg.read_image_bgr(wout)[..., 0]
    # oracle: per-frame full-plane extract with the stored factors (incl. the reference's [:L,:L]
    # truncation on this non-square plane), mean over frames, unscramble, min-max normalise
    est = np.mean([o.extract_plane(got[2 * j][0].astype(np.float32), data["Sc"][j], data["Uw"], data["Vwt"],
                                   0.15, 0.6, 64, 96, None) for j in range(5)], axis=0)
    want = np.clip(o.normalize_minmax(o.unpermute(est.astype(np.float32), idx)), 0, 255).astype(np.uint8)
    assert np.abs(ex.astype(int) - want.astype(int)).max() <= 2
    assert np.corrcoef(ex.ravel().astype(float), want.ravel().astype(float))[0, 1] > 0.999
    with pytest.raises(ValueError):
        v.embed_watermark_video(p, wp, outp, meta, password="pw", tile=4)
    # batched detect == per-frame detect
    marked = np.stack([g[0] for g in got[::2]])
    one = [gpu_ctx.ref_detect(marked[i], data["Sc"][i], data["Sw"], 0.15) for i in range(5)]
    assert np.allclose(gpu_ctx.ref_detect_planes(marked, data["Sc"], data["Sw"], 0.15), one, atol=1e-6)


@pytest.mark.gpu
def test_video_parameter_sweep(tmp_path, gpu_ctx):
    """Frame counts that do not divide the batch, frame_interval > 1, ragged frame sizes, one-frame
    clips, mono clips, both modes: marked frames are detected, unmarked clips are not, untouched
    frames and chroma stay bit-identical, the meta counts add up."""
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(31)
    wm = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)
    wp = str(tmp_path / "wm.png"); assert hg.write_png(wp, wm)
    cases = [dict(n=7, H=64, W=96, fi=1, batch=3, tile=8), dict(n=5, H=52, W=84, fi=2, batch=2, tile=8),
             dict(n=1, H=64, W=64, fi=1, batch=4, tile=8), dict(n=6, H=48, W=80, fi=3, batch=8, tile=None),
             dict(n=4, H=96, W=64, fi=1, batch=3, tile=None)]
    for i, c in enumerate(cases):
        sub = tmp_path / f"c{i}"; sub.mkdir()
        v, p, ys, chroma = _video(sub, n=c["n"], H=c["H"], W=c["W"], seed=40 + i)
        outp, meta, ps = v.embed_watermark_video(p, wp, str(sub / "out.y4m"), str(sub / "m.npz"), alpha=0.15,
                                                 frame_interval=c["fi"], password="pw", nonce=bytes(8),
                                                 batch=c["batch"], tile=c["tile"])
        data = np.load(meta, allow_pickle=False)
        n_marked = len(range(0, c["n"], c["fi"]))
        assert data["Sc"].shape[0] == n_marked and int(data["n_frames"]) == c["n"], c
        vid = v.Y4M(outp); got = [(y.copy(), ch.copy()) for _, y, ch in vid]; vid.close()
        assert len(got) == c["n"]
        for k, (y, ch) in enumerate(got):
            assert np.array_equal(ch, chroma[k])
            if k % c["fi"]:
                assert np.array_equal(y, ys[k])
            else:
                assert not np.array_equal(y, ys[k])
        ok, mean, scores = v.detect_watermark_video(outp, meta, batch=c["batch"])
        assert ok and scores.shape == (n_marked,) and scores.min() > 0.8, (c, scores)
        ok0, _, _ = v.detect_watermark_video(p, meta, batch=c["batch"])
        assert not ok0, c
        w = v.extract_watermark_video(outp, meta, str(sub / "w.png"), password="pw", batch=c["batch"])
        assert hg.read_image_bgr(w).shape[:2] == (c["H"], c["W"])


def _video444(tmp_path, n=5, H=64, W=96, seed=8):
    rng = np.random.default_rng(seed)
    # a smooth colourful clip (random chroma would leave the BGR gamut and be clipped by the conversion)
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for i in range(n):
        b = 90 + 60 * np.sin((xx + 3 * i) / 11.0) + rng.normal(0, 6, (H, W))
        g = 120 + 50 * np.cos((yy - 2 * i) / 7.0) + rng.normal(0, 6, (H, W))
        r = 100 + 40 * np.sin((xx + yy) / 13.0) + rng.normal(0, 6, (H, W))
        frames.append(np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8))
    ycc = [o.bgr_to_ycrcb(f) for f in frames]
    ys = np.stack([f[..., 0] for f in ycc])
    chroma = np.stack([np.concatenate([f[..., 2].ravel(), f[..., 1].ravel()]) for f in ycc])     # Cb plane, then Cr plane
    v = importlib.import_module(PKG_NAME + ".video")
    p = str(tmp_path / "in444.y4m")
    v.write_y4m(p, ys, chroma, chroma_tag="444")
    return v, p, ys, chroma


def _bgr_of(y, chroma):
    H, W = y.shape
    cb = chroma[:H * W].reshape(H, W); cr = chroma[H * W:].reshape(H, W)
    return o.ycrcb_to_bgr(np.stack([y, cr, cb], axis=-1))


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [8, None])
def test_video_color_444(tmp_path, gpu_ctx, tile):
    """Colour video on a 4:4:4 .y4m: every marked frame's B, G, R planes carry the colour watermark's B, G, R planes (decomposed
    once), held per plane against the oracle; the file round trip goes through the YCrCb conversion and is checked through it."""
    v, p, ys, chroma = _video444(tmp_path)
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    wm = np.random.default_rng(5).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    wp = str(tmp_path / "wmc.png"); assert hg.write_png(wp, wm)
    H, W = 64, 96
    key = o.derive_key("pw", bytes(8)); idx = o.permutation(H, W, o.rng_from_key(key))
    wm_r = o.resize_area(wm, W, H)
    w_s = [o.permute(wm_r[..., c].astype(np.float32), idx) for c in range(3)]
    wm_svd = [o.watermark_decompose(w, tile) for w in w_s]
    # array level: per plane against the oracle, bit-level bars of the grey path
    Uw, Sw, Vwt, idx_g = v.prepare_watermark_color(gpu_ctx, wm, H, W, key, tile)
    assert np.array_equal(idx_g, idx)
    for c in range(3):
        assert np.max(np.abs(Sw[c] - wm_svd[c][1])) / np.max(wm_svd[c][1]) < 1e-5
    bgr = np.stack([np.moveaxis(_bgr_of(ys[i], chroma[i]), -1, 0) for i in range(5)])               # [5, 3, H, W]
    K = 8 if tile else int(0.6 * 64)
    st, sc = v.embed_frames_color(gpu_ctx, bgr, Sw, 0.15, K, batch=2, tile=tile)
    for i in range(5):
        for c in range(3):
            ref = o.embed_plane(bgr[i, c].astype(np.float32), w_s[c], 0.15, 0.6, tile, wm_svd=wm_svd[c] if tile else None)
            assert np.abs(st[i, c].astype(int) - ref["stego"].astype(int)).max() <= 1
            assert np.mean(st[i, c] != ref["stego"]) < 5e-3
            assert np.max(np.abs(sc[c][i] - ref["Sc"])) / np.max(ref["Sc"]) < 1e-4
    # file level
    outp, meta, ps = v.embed_watermark_video_color(p, wp, str(tmp_path / "outc.y4m"), str(tmp_path / "vmc.npz"), alpha=0.15,
                                                   frame_interval=2, password="pw", nonce=bytes(8), batch=2, tile=tile)
    assert 15 < ps < 60
    data = np.load(meta, allow_pickle=False)
    assert str(data["mode"]) == "video_color" and int(data["n_frames"]) == 5
    for n in "bgr":
        assert data["S" + n].shape[0] == 3 and np.array_equal(data["S" + n], sc["bgr".index(n)][::2])
    vid = v.Y4M(outp); got = [(y.copy(), c.copy()) for _, y, c in vid]; vid.close()
    assert vid.chroma == "444" and len(got) == 5
    for i, (y, c) in enumerate(got):
        if i % 2:
            assert np.array_equal(y, ys[i]) and np.array_equal(c, chroma[i])                        # unmarked frames untouched
        else:
            want = o.bgr_to_ycrcb(np.moveaxis(st[i], 0, -1))                                        # the container's conversion
            assert np.array_equal(y, want[..., 0])
            assert np.array_equal(c, np.concatenate([want[..., 2].ravel(), want[..., 1].ravel()]))
    ok, mean, scores = v.detect_watermark_video_color(outp, meta)
    assert ok and mean > 0.9 and scores.shape == (3,)
    ok0, mean0, _ = v.detect_watermark_video_color(p, meta)
    assert not ok0 and mean0 < mean
    wout = v.extract_watermark_video_color(outp, meta, str(tmp_path / "wc.png"), password="pw")
    ex = hg.read_image_bgr(wout)
    assert ex.shape == (H, W, 3)
    for c in range(3):
        assert np.corrcoef(ex[..., c].ravel().astype(float), wm_r[..., c].ravel().astype(float))[0, 1] > 0.6
    with pytest.raises(ValueError, match="Sai mật khẩu"):
        v.extract_watermark_video_color(outp, meta, str(tmp_path / "x.png"), password="nope")
    # containers without full-resolution chroma are refused, and the grey reader refuses colour metadata
    v2, p420, _, _ = _video(tmp_path)
    with pytest.raises(ValueError, match="4:4:4"):
        v.embed_watermark_video_color(p420, wp, str(tmp_path / "o2.y4m"), str(tmp_path / "m2.npz"), password="pw")
    with pytest.raises(ValueError):
        v.detect_watermark_video(outp, meta)
