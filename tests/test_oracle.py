"""CPU tests of the oracle itself (no GPU).  The reference holds no golden
vectors, so what can be pinned independently is pinned here: the DCT against
its closed form, the SVD by reconstruction, the key / permutation / HMAC glue
against hashlib / RFC 4231 known answers, and the committed (oracle-generated)
fixtures as a drift check."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import wm_oracle as o

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("shape", [(8, 8), (16, 24), (64, 96)])
def test_dct_matches_closed_form(shape):
    x = np.random.default_rng(0).uniform(0, 255, shape).astype(np.float32)
    D1, D2 = o.dct_basis(shape[0]), o.dct_basis(shape[1])
    C = D1 @ x.astype(np.float64) @ D2.T
    assert np.abs(o.dct2(x) - C).max() < 1e-5 * np.abs(C).max()
    assert np.abs(o.idct2(o.dct2(x)) - x).max() < 1e-3
    I = D1 @ D1.T
    assert np.abs(I - np.eye(shape[0])).max() < 1e-12          # orthonormal: the DCT cancels in U S' V^T


def test_tile_dct_equals_plane_dct_of_each_tile():
    x = np.random.default_rng(1).uniform(0, 255, (16, 24)).astype(np.float32)
    T = o.to_tiles(x)
    Ct = o._dct_tiles(T)
    for ty in range(2):
        for tx in range(3):
            assert np.abs(Ct[ty, tx] - o.dct2(x[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8])).max() < 1e-3
    y = np.zeros_like(x); o.from_tiles(T, y)
    assert np.array_equal(x, y)


def test_svd_reconstruction_and_order():
    C = o.dct2(np.random.default_rng(2).uniform(0, 255, (32, 48)).astype(np.float32))
    U, S, Vt = o.svd_f32(C)
    assert U.dtype == S.dtype == Vt.dtype == np.float32          # fp64 dgesdd, cast back
    assert np.all(np.diff(S) <= 0)
    assert np.abs(U @ np.diag(S) @ Vt - C).max() < 1e-2
    assert np.abs(o.sigma_f32(C) - S).max() == 0


def test_k_formula():
    assert o.k_of(512, 0.6) == 307 and o.k_of(1080, 0.6) == 648
    assert o.k_of(8, 0.6) == 8 and o.k_of(8, 1.0) == 8                # degenerate at tile=8
    assert o.k_of(8, 0.0, k_floor=3) == 3 and o.k_of(10, 0.5, k_floor=1) == 5


@pytest.mark.parametrize("tile", [None, 8])
def test_alpha_zero_is_identity_up_to_truncation(tile):
    host = np.random.default_rng(3).integers(0, 256, (32, 32), dtype=np.uint8)
    wys = np.random.default_rng(4).integers(0, 256, (32, 32)).astype(np.float32)
    e = o.embed_plane(host.astype(np.float32), wys, 0.0, 0.6, tile)
    d = host.astype(int) - e["stego"].astype(int)
    assert d.min() >= 0 and d.max() <= 1                                # truncation, never +1
    assert np.abs(e["Yw"] - host).max() < 1e-2


@pytest.mark.parametrize("tile,shape", [(None, (32, 32)), (8, (32, 48)), (8, (45, 70))])
def test_extract_recovers_watermark_from_unquantised_stego(tile, shape):
    H, W = shape
    host = np.random.default_rng(5).integers(0, 256, shape, dtype=np.uint8)
    wys = np.random.default_rng(6).integers(0, 256, shape).astype(np.float32)
    e = o.embed_plane(host.astype(np.float32), wys, 0.15, 1.0, tile)
    w = o.extract_plane(e["Yw"], e["Sc"], e["Uw"], e["Vwt"], 0.15, 1.0, H, W, tile)
    Hb, Wb = (H, W) if tile is None else (H // 8 * 8, W // 8 * 8)
    assert np.abs(w[:Hb, :Wb] - wys[:Hb, :Wb]).max() < 0.5
    assert np.all(w[Hb:, :] == 0) and np.all(w[:, Wb:] == 0)
    assert o.detect_plane(e["Yw"], e["Sc"], e["Sw"], 0.15, tile) > 0.999
    if tile == 8:   # ragged border passes through embed untouched
        assert np.array_equal(e["stego"][Hb:, :], host[Hb:, :]) and np.array_equal(e["stego"][:, Wb:], host[:, Wb:])


def test_nonsquare_quirk_is_reproduced():
    """single:214-217: Uw[:L,:L] @ diag @ Vwt[:L,:L] drops columns L..W-1."""
    H, W = 16, 24
    host = np.random.default_rng(7).integers(0, 256, (H, W), dtype=np.uint8).astype(np.float32)
    wys = np.random.default_rng(8).integers(0, 256, (H, W)).astype(np.float32)
    e = o.embed_plane(host, wys, 0.15, 1.0, None)
    w = o.extract_plane(e["Yw"], e["Sc"], e["Uw"], e["Vwt"], 0.15, 1.0, H, W, None)
    S_cw = o.sigma_f32(o.dct2(e["Yw"]))
    sh = (S_cw - e["Sc"]) / 0.15
    full = np.zeros((H, W), np.float32)
    full[:16, :16] = (e["Uw"][:16, :16] @ np.diag(sh) @ e["Vwt"][:16, :16]).astype(np.float32)
    assert np.abs(w - o.idct2(full)).max() < 1e-3
    assert np.abs(w - wys).max() > 10                                   # hence NOT the true watermark


def test_security_known_answers():
    assert o.derive_key("pw", b"\x00" * 8) == hashlib.sha256(b"pw" + b"\x00" * 8).digest()
    # RFC 4231 test case 1
    assert o.hmac_digest(b"\x0b" * 20, [b"Hi ", b"There"]).hex() == \
        "b0344c61d8db38535ca8afceaf0bf12b881dc200c9833da726e9376c2e32cff7"
    key = o.derive_key("bench", bytes(8))
    idx = o.permutation(4, 6, o.rng_from_key(key))
    assert sorted(idx.tolist()) == list(range(24))
    x = np.arange(24, dtype=np.float32).reshape(4, 6)
    assert np.array_equal(o.unpermute(o.permute(x, idx), idx), x)
    idx2 = o.permutation(4, 6, o.rng_from_key(key))
    assert np.array_equal(idx, idx2)
    seed = int.from_bytes(key[:8], "big")
    ref = np.arange(24); np.random.default_rng(seed).shuffle(ref)
    assert np.array_equal(idx, ref)


def test_metrics_and_colour_glue():
    a = np.random.default_rng(9).integers(0, 256, (32, 32, 3), dtype=np.uint8)
    assert o.psnr(a, a) == 99.0 and abs(o.ssim(a, a) - 1.0) < 1e-6
    b = a.copy(); b[0, 0, 0] ^= 0x10
    assert 40 < o.psnr(a, b) < 99
    g = o.bgr_to_gray(np.full((2, 2, 3), 200, np.uint8))
    assert np.all(g == 200)
    ycc = o.bgr_to_ycrcb(a)
    assert np.abs(o.ycrcb_to_bgr(ycc).astype(int) - a.astype(int)).max() <= 3     # 8-bit fixed point round trip
    small = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert np.array_equal(o.resize_area(small, 8, 6), np.kron(small, np.ones((2, 2), np.uint8)))
    assert np.array_equal(o.resize_area(np.kron(small, np.ones((2, 2), np.uint8)), 4, 3), small)
    n = o.normalize_minmax(np.array([[1.0, 3.0], [5.0, 9.0]], np.float32))
    assert n.min() == 0 and abs(n.max() - 255) < 1e-3


def test_password_errors():
    cover = np.zeros((16, 16, 3), np.uint8); wm = np.zeros((4, 4, 3), np.uint8)
    with pytest.raises(ValueError):
        o.embed_arrays(cover, wm, "", bytes(8))
    r = o.embed_arrays(np.random.default_rng(1).integers(0, 256, (16, 16, 3), dtype=np.uint8),
                       np.random.default_rng(2).integers(0, 256, (4, 4, 3), dtype=np.uint8), "pw", bytes(8), tile=8)
    with pytest.raises(ValueError, match="Sai mật khẩu"):
        o.extract_arrays(r["stego"], r["meta"], "other", tile=8)
    with pytest.raises(ValueError):
        o.extract_arrays(r["stego"], r["meta"], "", tile=8)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden_fixture(path):
    """Drift check: fixtures are oracle-generated (tests/golden/make_golden.py).
    LAPACK builds may differ in the last bit, so singular values / scores are
    compared to 1e-5 and pixels to 1 LSB, not byte-for-byte."""
    g = np.load(path, allow_pickle=False)
    tile = None if int(g["tile"]) < 0 else int(g["tile"])
    nonce = bytes(g["meta_nonce"].tolist())
    r = o.embed_arrays(g["cover"], g["wm"], "golden-pw", nonce, float(g["alpha"]), bool(g["color"]),
                       float(g["kfrac"]), tile, int(g["k_floor"]))
    assert np.abs(r["stego"].astype(int) - g["stego"].astype(int)).max() <= 1
    assert np.mean(r["stego"] != g["stego"]) < 1e-3
    assert abs(r["psnr"] - float(g["psnr"])) < 1e-2 and abs(r["ssim"] - float(g["ssim"])) < 1e-4
    for k in ("Sc", "Sw", "Sb", "SWb"):
        if "meta_" + k in g:
            a, b = r["meta"][k], g["meta_" + k]
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    key = o.derive_key("golden-pw", nonce)
    H, W = g["cover"].shape[:2]
    idx = o.permutation(H, W, o.rng_from_key(key))
    assert hashlib.sha256(idx.astype(np.int64).tobytes()).digest() == bytes(g["perm_sha256"].tolist())
    ok, score = o.detect_arrays(g["stego"], r["meta"], 0.6, tile)
    assert abs(score - float(g["detect_score"])) < 1e-4
    ex = o.extract_arrays(g["stego"], r["meta"], "golden-pw", True, tile, int(g["k_floor"]))
    assert np.mean(np.abs(ex.astype(int) - g["extracted"].astype(int)) > 1) < 1e-2


def test_dct_against_the_published_jpeg_worked_example():
    """An EXTERNAL known answer for the 2-D orthonormal DCT-II (what cv2.dct computes, single:32-33): the 8x8 worked
    example of the JPEG literature (level-shifted sample block -> DCT coefficients printed to two decimals; e.g. the
    'JPEG' article of Wikipedia, DC term -415.38), stored in tests/golden/external/jpeg_dct_example.npz.  Not held by the
    reference (it holds nothing), so parity stays 'unpinned' - but it is a vector neither the oracle nor the kernels
    produced."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "external", "jpeg_dct_example.npz"))
    X, C = g["block"], g["dct"]
    assert abs(C[0, 0] + 415.38) < 1e-9
    assert np.abs(o.dct2(X.astype(np.float32)) - C).max() < 6e-3          # two printed decimals
    assert np.abs(o.idct2(o.dct2(X.astype(np.float32))) - X).max() < 1e-3
    D = o.dct_basis(8)
    assert np.abs(D @ X @ D.T - C).max() < 6e-3


def _wikipedia_svd_tile():
    """The 4 x 5 matrix of Wikipedia's 'Singular value decomposition' article (singular values 3, sqrt(5), 2, 0),
    zero-padded to an 8 x 8 uint8 tile: an external known answer for the tile SVD."""
    m = np.zeros((8, 8), np.uint8)
    m[:4, :5] = [[1, 0, 0, 0, 2], [0, 0, 3, 0, 0], [0, 0, 0, 0, 0], [0, 2, 0, 0, 0]]
    return m, np.array([3.0, np.sqrt(5.0), 2.0, 0, 0, 0, 0, 0])


def test_tile_singular_values_against_a_published_example():
    m, want = _wikipedia_svd_tile()
    s = o.stego_sigma(m.astype(np.float32), 8).reshape(-1)
    assert np.abs(s - want).max() < 1e-5
