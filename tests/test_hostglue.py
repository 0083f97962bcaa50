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
    for (W, H) in ((112, 80), (28, 20), (50, 33), (56, 40)):
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
