"""GPU parity: HIP tile-mode kernels (through the C ABI) vs the NumPy oracle.

Tolerances (BASELINE.json north_star): singular values within 1e-4 relative
(to sigma_1 of the tile), uint8 stego within 1 LSB."""
import numpy as np
import pytest

from oracle import wm_oracle as o

pytestmark = pytest.mark.gpu

SIGMA_RTOL = 1e-4


def _inputs(H, W, seed=1234, wseed=4321):
    host = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8)
    wm = np.random.default_rng(wseed).integers(0, 256, (H, W), dtype=np.uint8)
    key = o.derive_key("bench", bytes(8))
    idx = o.permutation(H, W, o.rng_from_key(key))
    return host, o.permute(wm.astype(np.float32), idx)


def _rel_sigma(a, b):
    a = a.reshape(-1, 8); b = b.reshape(-1, 8)
    return float(np.max(np.abs(a - b) / np.maximum(b[:, :1], 1e-30)))


@pytest.mark.parametrize("H,W", [(8, 8), (64, 96), (512, 512), (1080, 1920)])
def test_embed_parity(gpu_ctx, H, W):
    alpha = 0.15
    host, wys = _inputs(H, W)
    ref = o.embed_plane(host.astype(np.float32), wys, alpha, 0.6, tile=8)
    stego, sc, yw = gpu_ctx.embed_tiles(host, ref["Sw"], alpha, K=8, want_yw=True)
    assert _rel_sigma(sc, ref["Sc"]) < SIGMA_RTOL
    d = np.abs(stego.astype(np.int32) - ref["stego"].astype(np.int32))
    assert d.max() <= 1
    assert np.mean(d != 0) < 2e-3          # truncation flips only where Yw sits on an integer
    assert np.abs(yw - ref["Yw"]).max() < 3e-2   # (w_i/s_i) * eps * |X| for the smallest s_i
    assert abs(o.psnr(host, stego) - o.psnr(host, ref["stego"])) < 1e-3


def test_svd_tiles_parity(gpu_ctx):
    H, W = 256, 320
    _, wys = _inputs(H, W)
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    Uo, So, Vto = o.watermark_decompose(wys, 8)
    assert _rel_sigma(S, So) < SIGMA_RTOL
    # singular vectors are sign-ambiguous: compare the rank-1 terms
    rec = np.matmul(U * S[..., None, :], Vt)
    reco = np.matmul(Uo * So[..., None, :], Vto)
    assert np.abs(rec - reco).max() < 5e-3
    I = np.eye(8, dtype=np.float32)
    assert np.abs(np.matmul(U.swapaxes(-1, -2), U) - I).max() < 1e-5
    assert np.abs(np.matmul(Vt, Vt.swapaxes(-1, -2)) - I).max() < 1e-5
    assert np.all(np.diff(S, axis=-1) <= 1e-3 * S[..., :1])   # descending like LAPACK


@pytest.mark.parametrize("K", [8, 3])
def test_sigma_extract_detect_parity(gpu_ctx, K):
    H, W, alpha = 512, 512, 0.15
    host, wys = _inputs(H, W)
    ref = o.embed_plane(host.astype(np.float32), wys, alpha, kfrac=0.0, tile=8, k_floor=K)
    st = ref["stego"]
    s = gpu_ctx.sigma_tiles(st)
    assert _rel_sigma(s, o.stego_sigma(st.astype(np.float32), 8)) < SIGMA_RTOL
    w = gpu_ctx.extract_tiles(st, ref["Sc"], ref["Uw"], ref["Vwt"], alpha, K=K)
    wo = o.extract_plane(st.astype(np.float32), ref["Sc"], ref["Uw"], ref["Vwt"], alpha, 0.0, H, W, 8, k_floor=K)
    assert np.abs(w - wo).max() < 2e-2          # values span ~[-300, 300]
    score = gpu_ctx.detect_tiles(st, ref["Sc"], ref["Sw"], alpha)[0]
    assert abs(score - o.detect_plane(st.astype(np.float32), ref["Sc"], ref["Sw"], alpha, 8)) < 1e-4
    assert score > 0.9
    # an unrelated image scores low, and the same as the oracle says
    other = np.random.default_rng(99).integers(0, 256, (H, W), dtype=np.uint8)
    s_other = gpu_ctx.detect_tiles(other, ref["Sc"], ref["Sw"], alpha)[0]
    assert abs(s_other - o.detect_plane(other.astype(np.float32), ref["Sc"], ref["Sw"], alpha, 8)) < 1e-4
    assert s_other < 0.6


def test_ragged_and_batched(gpu_ctx):
    """H, W not multiples of 8: border passes through embed and reads 0 in
    extract; batched planes with a shared sigma_w."""
    H, W, alpha = 45, 70, 0.12
    rng = np.random.default_rng(7)
    hosts = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    wys = rng.integers(0, 256, (H, W)).astype(np.float32)
    Uo, So, Vto = o.watermark_decompose(wys, 8)
    stego, sc, yw = gpu_ctx.embed_tiles(hosts, So, alpha, want_yw=True)
    for p in range(3):
        ref = o.embed_plane(hosts[p].astype(np.float32), wys, alpha, 0.6, 8, wm_svd=(Uo, So, Vto))
        assert np.abs(stego[p].astype(int) - ref["stego"].astype(int)).max() <= 1
        assert np.array_equal(stego[p][40:, :], hosts[p][40:, :])
        assert np.array_equal(stego[p][:, 64:], hosts[p][:, 64:])
        assert _rel_sigma(sc[p], ref["Sc"]) < SIGMA_RTOL
    w = gpu_ctx.extract_tiles(stego, sc, Uo, Vto, alpha)
    assert np.all(w[:, 40:, :] == 0) and np.all(w[:, :, 64:] == 0)
    wo = o.extract_plane(stego[1].astype(np.float32), sc[1], Uo, Vto, alpha, 0.6, H, W, 8)
    assert np.abs(w[1] - wo).max() < 2e-2


def test_empty_and_bad_args(gpu_ctx):
    z = np.zeros((0, 0), np.uint8)
    st, sc, _ = gpu_ctx.embed_tiles(z, np.zeros((0, 0, 8), np.float32), 0.1)
    assert st.shape == (0, 0)
    small = np.full((5, 7), 9, np.uint8)          # no full tile: pure pass-through
    st, sc, _ = gpu_ctx.embed_tiles(small, np.zeros((0, 0, 8), np.float32), 0.1)
    assert np.array_equal(st, small) and sc.size == 0
    assert gpu_ctx.detect_tiles(small, np.zeros((0, 0, 8), np.float32), np.zeros((0, 0, 8), np.float32), 0.1)[0] == 0.0
    with pytest.raises(ValueError):
        gpu_ctx.embed_tiles(np.zeros((8, 8), np.uint8), np.zeros((1, 1, 8), np.float32), 0.1, K=9)
    with pytest.raises(ValueError):
        gpu_ctx.embed_tiles(np.zeros((8, 8), np.float32), np.zeros((1, 1, 8), np.float32), 0.1)


def test_rank_deficient_tiles_on_gpu(gpu_ctx):
    """Flat / saturated / rank-1 / rank-2 tiles go through the fallback kernel's
    orthonormal completion: same property checks as the CPU build of the math."""
    from test_host_harness import _degenerate_image, check_completion_properties
    img, mask = _degenerate_image()
    H, W = img.shape
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    ref = o.embed_plane(img.astype(np.float32), wys, 0.15, 0.6, 8)
    stego, sc, yw = gpu_ctx.embed_tiles(img, ref["Sw"], 0.15, want_yw=True)
    check_completion_properties(img, mask, wys, 0.15, stego, sc, yw, ref)
    # all-flat plane: every tile takes the fallback path
    flat = np.full((64, 64), 77, np.uint8)
    st, sc2, yw2 = gpu_ctx.embed_tiles(flat, ref["Sw"][:8, :8], 0.15, want_yw=True)
    assert np.isfinite(yw2).all() and abs(float(sc2[0, 0, 0]) - 77 * 8) < 1e-2
    # watermark side: degenerate plane keeps orthonormal factors
    wflat = np.zeros((16, 16), np.float32); wflat[:8, :8] = 255; wflat[8:, 8:] = np.arange(8)[None, :]
    U, S, Vt = gpu_ctx.svd_tiles(wflat)
    I = np.eye(8, dtype=np.float32)
    assert np.abs(np.matmul(U.swapaxes(-1, -2), U) - I).max() < 1e-5
    assert np.abs(np.matmul(Vt, Vt.swapaxes(-1, -2)) - I).max() < 1e-5


def test_constant_tiles_take_the_closed_form(gpu_ctx):
    """Letterbox bars / flat areas: constant tiles are finished in closed form (wm::embed_tile_constant) whether a
    whole wave is constant (the fast kernel skips its iteration) or they sit next to textured or rank-deficient
    tiles.  The device result equals the CPU build of the same arithmetic bit for bit on those tiles (pure FMA chains
    over tabulated constants), the reference's invariant svd(Yw) = Sc + alpha Sw holds, the rest of the plane is
    unchanged by their presence, and in-place embedding gives the same plane."""
    import ctypes as C
    import __graft_entry__ as ge
    hh = C.CDLL(ge.build_host_harness())
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(77)
    H, W = 128, 1024                                             # 128 tiles per row: two waves per tile row
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[:24] = 16; img[-24:] = 0                                 # bars: whole waves of constant tiles (studio and full-range black)
    img[40:56, 100:300] = 255                                    # a flat patch inside texture: mixed waves
    img[64:72, 512:] = np.arange(8, dtype=np.uint8)[:, None] * 30    # rank-1, not constant, next to ...
    img[64:72, 600:700] = 90                                     # ... constant tiles in the same wave
    nby, nbx = H // 8, W // 8
    sw = np.sort(rng.uniform(1, 1500, (nby, nbx, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
    stego, sc, yw = gpu_ctx.embed_tiles(img, sw, 0.15, want_yw=True)
    st_h = np.empty_like(img); sc_h = np.empty((nby * nbx, 8), np.float32); yw_h = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    hh.hh_embed_tiles_u8_pk(vp(img), vp(sw), vp(st_h), vp(sc_h), vp(yw_h), H, W, W, C.c_float(0.15), 8, C.byref(ms), C.byref(nf))
    tiles = img.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
    const = (tiles == tiles[:, :, :1, :1]).all(axis=(2, 3))
    assert const.sum() == 6 * nbx + 2 * 24 + 12               # bars + the tiles fully inside the two flat patches
    T = lambda x: x.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
    assert np.array_equal(T(yw)[const], T(yw_h)[const])          # closed form: the same FMA chain on both sides
    assert np.array_equal(sc[const], sc_h.reshape(nby, nbx, 8)[const])
    # full-rank tiles: GPU vs CPU build as everywhere (rank-deficient non-constant ones - the rank-1 strip, tiles cut by a
    # patch edge - go through the literal completion, whose directions are only defined as a set)
    full = np.linalg.matrix_rank(tiles.astype(np.float64)) == 8
    d = np.abs(T(stego).astype(int) - T(st_h).astype(int))[full]
    assert full.sum() > 1000 and d.max() <= 1 and np.mean(d != 0) < 1e-3
    got = np.linalg.svd(T(yw)[const].astype(np.float64), compute_uv=False)
    want = np.sort(sc[const].astype(np.float64) + 0.15 * sw[const], axis=-1)[:, ::-1]
    assert np.max(np.abs(got - want) / np.maximum(want[:, :1], 1.0)) < 1e-4
    assert np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8))
    # the textured tiles do not care what the bars hold
    img2 = img.copy(); img2[:24] = rng.integers(0, 256, (24, W), dtype=np.uint8)
    stego2, _, _ = gpu_ctx.embed_tiles(img2, sw, 0.15)
    assert np.array_equal(stego2[24:], stego[24:])
    # in place (stego aliases host): the constant tiles are still read after the fast kernel has written its tiles
    buf = img.copy(); sc_ip = np.empty((nby * nbx, 8), np.float32)
    cp = lambda a_: C.c_void_p(a_.ctypes.data)
    gpu_ctx._call("wm_embed_tiles_u8", cp(buf), cp(sw), cp(buf), cp(sc_ip), None, 1, H, W, W, H * W, 0, 0.15, 8)
    assert np.array_equal(buf, stego) and np.array_equal(sc_ip.reshape(sc.shape), sc)


def test_rank1_tiles_take_the_closed_form(gpu_ctx):
    """Screen-like content: rank-1 tiles - rows all equal (vertical edges and rules), columns all equal (horizontal
    ones), rule crossings on a flat background - are finished by wm::embed_tile_rank1 (third list of the fast kernel)
    whether a whole wave is made of such tiles and constant ones (the iteration is skipped) or they sit among textured tiles.  Device = CPU build of the same
    arithmetic up to the reciprocal square root (v_rsq_f32 against 1 / sqrtf: Yw within 1e-3), Sc = (sigma_1, 0, ..) with
    exact zeros, the reference's invariant svd(Yw) = Sc + alpha Sw, independence from the other tiles, in-place embedding."""
    import ctypes as C
    import __graft_entry__ as ge
    hh = C.CDLL(ge.build_host_harness())
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(78)
    H, W = 128, 1024                                             # 128 tiles per row: two waves per tile row
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[:32] = 240                                               # flat background ...
    img[:32, 100:103] = 0; img[:32, 517] = 30                    # ... with vertical rules: rows-equal tiles among constant ones (whole waves)
    img[12] = 0; img[27, :512] = 240 - 90                        # horizontal rules: columns-equal tiles; the crossings with the black rule are rank 1 too
    img[64:72, 256:512] = np.repeat(rng.integers(0, 256, (1, 256), dtype=np.uint8), 8, axis=0)    # rows-equal strip inside texture (mixed waves)
    img[80:96, 600:640] = np.repeat(rng.integers(0, 256, (16, 1), dtype=np.uint8), 40, axis=1)    # columns-equal patch inside texture
    nby, nbx = H // 8, W // 8
    sw = np.sort(rng.uniform(1, 1500, (nby, nbx, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
    stego, sc, yw = gpu_ctx.embed_tiles(img, sw, 0.15, want_yw=True)
    st_h = np.empty_like(img); sc_h = np.empty((nby * nbx, 8), np.float32); yw_h = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    hh.hh_embed_tiles_u8_pk(vp(img), vp(sw), vp(st_h), vp(sc_h), vp(yw_h), H, W, W, C.c_float(0.15), 8, C.byref(ms), C.byref(nf))
    T = lambda x: x.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
    tiles = T(img)
    rows_eq = (tiles == tiles[:, :, :1, :]).all(axis=(2, 3)); cols_eq = (tiles == tiles[:, :, :, :1]).all(axis=(2, 3))
    rk = np.linalg.matrix_rank(tiles.astype(np.float64))
    r1 = (rk == 1) & ~(rows_eq & cols_eq)
    assert rows_eq[8].sum() == 32 and cols_eq[10:12, 75:80].all() and r1.sum() > 150
    assert (r1 & ~rows_eq & ~cols_eq).sum() >= 1                 # a crossing: rank 1 with neither rows nor columns equal
    assert np.abs(T(yw)[r1] - T(yw_h)[r1]).max() < 1e-3
    assert np.abs(sc[r1][:, 0] - sc_h.reshape(nby, nbx, 8)[r1][:, 0]).max() < 1e-3 and not sc[r1][:, 1:].any()
    X = tiles[r1].astype(np.float64)
    assert np.abs(sc[r1][:, 0] - np.linalg.svd(X, compute_uv=False)[:, 0]).max() < 1e-3
    got = np.linalg.svd(T(yw)[r1].astype(np.float64), compute_uv=False)
    want = np.sort(sc[r1].astype(np.float64) + 0.15 * sw[r1], axis=-1)[:, ::-1]
    assert np.max(np.abs(got - want) / np.maximum(want[:, :1], 1.0)) < 2e-5
    assert np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8))
    # every other tile: as in the CPU build (full-rank ones to 1 LSB; constant ones bit for bit)
    full = np.linalg.matrix_rank(tiles.astype(np.float64)) == 8
    d = np.abs(T(stego).astype(int) - T(st_h).astype(int))[full]
    assert full.sum() > 1000 and d.max() <= 1 and np.mean(d != 0) < 1e-3
    const = rows_eq & cols_eq
    assert const.sum() > 300 and np.array_equal(T(yw)[const], T(yw_h)[const])
    # the textured tiles do not care what the structured rows hold, and a rank-1 tile does not care about its neighbours
    img2 = img.copy(); img2[32:64] = rng.integers(0, 256, (32, W), dtype=np.uint8)
    stego2, _, yw2 = gpu_ctx.embed_tiles(img2, sw, 0.15, want_yw=True)
    assert np.array_equal(stego2[:32], stego[:32]) and np.array_equal(yw2[64:], yw[64:])
    # in place
    buf = img.copy(); sc_ip = np.empty((nby * nbx, 8), np.float32)
    cp = lambda a_: C.c_void_p(a_.ctypes.data)
    gpu_ctx._call("wm_embed_tiles_u8", cp(buf), cp(sw), cp(buf), cp(sc_ip), None, 1, H, W, W, H * W, 0, 0.15, 8)
    assert np.array_equal(buf, stego) and np.array_equal(sc_ip.reshape(sc.shape), sc)


def test_one_small_singular_value_tiles_on_gpu(gpu_ctx):
    """Kind 3 of the flagged-tile lists (k_embed_one_small): tiles whose eighth singular value is below 1e-5 sigma_1 - what
    noise frames contain, 4 in 10 000 - are completed from the fast kernel's own B, both eighth vectors as orthogonal
    complements, their joint sign from a float64 bilinear form.  Near-singular FULL-rank tiles have unique singular
    vectors: 1 LSB against float64 LAPACK (the literal chain they used to take was off by 3 LSB on the ones below
    float32 resolution: tools/one_small_check.py); rank-7 tiles satisfy the reference's invariant; a plane made of rank-7
    tiles only overflows the kind's sub-lists into the literal chain and still satisfies it."""
    from test_host_harness import _one_small_tiles
    near, r7 = _one_small_tiles(n_want=40, seed=9)
    rng = np.random.default_rng(3)
    H, W = 64, 1024
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    nby, nbx = H // 8, W // 8
    where = []
    for k, t in enumerate(near + r7):                              # scattered over the plane: mixed waves
        ty, tx = (7 * k) % nby, (13 * k + 5) % nbx
        img[ty * 8: ty * 8 + 8, tx * 8: tx * 8 + 8] = t
        where.append((ty, tx))
    assert len(set(where)) == len(where)
    sw = np.sort(rng.uniform(1, 1500, (nby, nbx, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
    stego, sc, yw = gpu_ctx.embed_tiles(img, sw, 0.15, want_yw=True)
    T = lambda x: x.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
    X = T(img).astype(np.float64)
    U, S, Vt = np.linalg.svd(X)
    ref = (U * (S + 0.15 * sw)[..., None, :]) @ Vt
    full = S[..., 7] > 1e-9 * S[..., 0]                            # everything but the exactly rank-7 tiles: unique vectors
    q = np.abs(np.clip(T(yw), 0, 255).astype(np.uint8).astype(int) - np.clip(ref, 0, 255).astype(np.uint8).astype(int))
    assert q[full].max() <= 1 and np.mean(q[full] != 0) < 1e-3
    near_idx = tuple(np.array(where[:len(near)]).T)
    assert (S[near_idx][:, 7] < 1e-5 * S[near_idx][:, 0]).all()
    assert np.abs(T(yw)[near_idx] - ref[near_idx]).max() < 0.05
    got = np.linalg.svd(T(yw).astype(np.float64), compute_uv=False)
    want = np.sort(sc.astype(np.float64) + 0.15 * sw, axis=-1)[..., ::-1]
    assert np.max(np.abs(got - want) / want[..., :1]) < 1e-4
    assert np.max(np.abs(sc - S) / S[..., :1]) < 2e-6
    assert np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8))
    # a plane of rank-7 tiles only (two equal rows in every tile): 256 waves on 64 sub-lists of 64 entries - most of the tiles
    # overflow into the literal chain
    big = rng.integers(0, 256, (1024, 1024), dtype=np.uint8)
    big[5::8] = big[2::8]
    swb = np.sort(rng.uniform(1, 1500, (128, 128, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
    st2, sc2, yw2 = gpu_ctx.embed_tiles(big, swb, 0.15, want_yw=True)
    T2 = yw2.reshape(128, 8, 128, 8).transpose(0, 2, 1, 3).astype(np.float64)
    got = np.linalg.svd(T2, compute_uv=False)
    want = np.sort(sc2.astype(np.float64) + 0.15 * swb, axis=-1)[..., ::-1]
    assert np.isfinite(yw2).all() and np.max(np.abs(got - want) / want[..., :1]) < 1e-4
    s_true = np.linalg.svd(big.reshape(128, 8, 128, 8).transpose(0, 2, 1, 3).astype(np.float64), compute_uv=False)
    assert np.max((np.abs(sc2 - s_true) - 4 * 2.0 ** -14) / s_true[..., :1]) < 1e-5
    gpu_ctx.check_status()


def _scene(rng, H, W):
    """UI-like / mixed content: flat areas, rectangles, rules, gradients along one or two axes, saturated patches, noise
    and camera-like texture - every kind of tile the embed kernels classify (constant, rank 1, rank 2 .. 7, full rank)."""
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.full((H, W), int(rng.integers(0, 256)), np.float64)
    for _ in range(int(rng.integers(5, 40))):
        y, x = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
        h, w = int(rng.integers(1, H // 2)), int(rng.integers(1, W // 2))
        kind = int(rng.integers(0, 7))
        sl = (slice(y, y + h), slice(x, x + w))
        if kind == 0: img[sl] = rng.integers(0, 256)
        elif kind == 1: img[sl] = rng.integers(0, 256, img[sl].shape)
        elif kind == 2: img[sl] = (xx[sl] * rng.uniform(0.1, 3)) % 256
        elif kind == 3: img[sl] = (yy[sl] * rng.uniform(0.1, 3)) % 256
        elif kind == 4: img[sl] = (xx[sl] * rng.uniform(0.1, 2) + yy[sl] * rng.uniform(0.1, 2)) % 256
        elif kind == 5: img[sl] = 128 + 60 * np.sin(xx[sl] / rng.uniform(3, 40)) * np.cos(yy[sl] / rng.uniform(3, 40)) + rng.normal(0, 1.5, img[sl].shape)
        else: hh_, ww_ = img[sl].shape; img[sl] = np.outer(rng.integers(0, 16, hh_), rng.integers(0, 16, ww_))   # products: rank-1 tiles without equal rows
    for _ in range(int(rng.integers(0, 6))):
        img[int(rng.integers(0, H))] = rng.integers(0, 256)
        img[:, int(rng.integers(0, W))] = rng.integers(0, 256)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_random_scenes_every_tile_class(gpu_ctx):
    """Property test over generated scenes: whatever path a tile takes (fast, constant, rank 1, completion from B, literal
    chain) the reference's guarantees hold - svd(Yw) = Sc + alpha Sw[:K], Sc = the tile's singular values, stego =
    clip(Yw) truncated - and tiles with unique singular vectors (s_8 > 1e-9 s_1, distinct values) match float64 LAPACK
    to 1 LSB."""
    rng = np.random.default_rng(2026)
    n_unique = 0
    for case in range(10):
        H, W = int(rng.choice([64, 128, 256])), int(rng.choice([128, 256, 512]))
        K = int(rng.choice([8, 8, 5, 1])); alpha = float(rng.uniform(0.05, 0.25))
        img = _scene(rng, H, W)
        nby, nbx = H // 8, W // 8
        sw = np.sort(rng.uniform(1, 1500, (nby, nbx, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
        stego, sc, yw = gpu_ctx.embed_tiles(img, sw, alpha, K=K, want_yw=True)
        T = lambda x: x.reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3)
        X = T(img).astype(np.float64)
        U, S, Vt = np.linalg.svd(X)
        w = alpha * sw.astype(np.float64); w[..., K:] = 0
        assert np.isfinite(yw).all() and np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8)), case
        assert np.max((np.abs(sc - S) - 4 * 2.0 ** -14) / np.maximum(S[..., :1], 1.0)) < 1e-5, case
        got = np.linalg.svd(T(yw).astype(np.float64), compute_uv=False)
        want = np.sort(sc.astype(np.float64) + w, axis=-1)[..., ::-1]
        # (a kept singular value within a decade of the 1e-5 s_1 cut has vectors good to ~eps s_1 / s_i = 1e-3: 5e-4 here,
        # 1e-4 everywhere else in this file)
        assert np.max(np.abs(got - want) / np.maximum(want[..., :1], 1.0)) < 5e-4, case
        gaps = np.min(-np.diff(S, axis=-1) / np.maximum(S[..., :1], 1e-30), axis=-1)
        unique = (S[..., 7] > 1e-9 * S[..., 0]) & (gaps > 1e-4)          # well separated: the singular vectors are defined
        ref = (U * (S + w)[..., None, :]) @ Vt
        q = np.abs(np.clip(T(yw), 0, 255).astype(np.uint8).astype(int) - np.clip(ref, 0, 255).astype(np.uint8).astype(int))
        assert not unique.any() or q[unique].max() <= 1, (case, int(q[unique].max()) if unique.any() else 0)
        n_unique += int(unique.sum())
        # the sigma-only kernels on the same content (extract / detect of structured images): no sweep-bound flag, values right
        for plane in (img, stego):
            sg = gpu_ctx.sigma_tiles(plane)
            sref = np.linalg.svd(T(plane).astype(np.float64), compute_uv=False)
            assert np.max(np.abs(sg - sref) / np.maximum(sref[..., :1], 1.0)) < 2e-5, case
        score = gpu_ctx.detect_tiles(stego, sc, sw, alpha)
        assert np.isfinite(score).all()
    assert n_unique > 500
    gpu_ctx.check_status()


def test_unaligned_and_strided_planes(gpu_ctx, hostapi):
    """Byte-wise kernel variants: row stride / base address not multiples of 8,
    planes embedded in a larger buffer (row_stride > W, plane_stride > H*row_stride)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    big = rng.integers(0, 256, (3, 70, 101), dtype=np.uint8)
    view = big[:, 3:3 + 48, 5:5 + 72]                      # base offset 5 (odd), row stride 101, plane stride 7070
    dense = np.ascontiguousarray(view)
    s_strided = gpu_ctx.sigma_tiles(view)
    s_dense = gpu_ctx.sigma_tiles(dense)
    assert np.array_equal(s_strided, s_dense)
    assert _rel_sigma(s_dense[1], o.stego_sigma(dense[1].astype(np.float32), 8)) < SIGMA_RTOL
    # embed through the raw ABI with strided input AND strided output; bytes outside the planes survive
    wys = rng.integers(0, 256, (48, 72)).astype(np.float32)
    Uo, So, Vto = o.watermark_decompose(wys, 8)
    sw = np.ascontiguousarray(So.reshape(-1, 8))
    out_big = np.full_like(big, 7)
    sc = np.empty((3, 6 * 9, 8), np.float32)
    vp = lambda a, off=0: C.c_void_p(a.ctypes.data + off)
    off = 3 * 101 + 5
    gpu_ctx._call("wm_embed_tiles_u8", vp(big, off), vp(sw), vp(out_big, off), vp(sc), None,
                  3, 48, 72, 101, 70 * 101, 0, 0.15, 8)
    for p in range(3):
        ref = o.embed_plane(dense[p].astype(np.float32), wys, 0.15, 0.6, 8, wm_svd=(Uo, So, Vto))
        got = out_big[p, 3:51, 5:77]
        assert np.abs(got.astype(int) - ref["stego"].astype(int)).max() <= 1
        assert _rel_sigma(sc[p], ref["Sc"]) < SIGMA_RTOL
    mask = np.ones_like(big, bool); mask[:, 3:51, 5:77] = False
    assert np.all(out_big[mask] == 7)


def test_in_place_embed_with_deficient_tiles(gpu_ctx):
    """stego may alias host (include/wmhip.h): flagged tiles must be embedded once, from
    the original pixels, by the fallback kernel."""
    import ctypes as C
    from test_host_harness import _degenerate_image
    img, _ = _degenerate_image()
    H, W = img.shape
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    Uo, So, Vto = o.watermark_decompose(wys, 8)
    sw = np.ascontiguousarray(So.reshape(-1, 8))
    want, sc_want, _ = gpu_ctx.embed_tiles(img, So, 0.15)
    buf = img.copy(); sc = np.empty((sw.shape[0], 8), np.float32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    gpu_ctx._call("wm_embed_tiles_u8", vp(buf), vp(sw), vp(buf), vp(sc), None, 1, H, W, W, H * W, 0, 0.15, 8)
    assert np.array_equal(buf, want) and np.allclose(sc.reshape(sc_want.shape), sc_want)


def test_results_are_bitwise_reproducible(gpu_ctx):
    """No float atomics anywhere in the numeric path: two runs give identical bytes."""
    host, wys = _inputs(256, 384)
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    a = gpu_ctx.embed_tiles(host, S, 0.15, want_yw=True)
    b = gpu_ctx.embed_tiles(host, S, 0.15, want_yw=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(gpu_ctx.extract_tiles(a[0], a[1], U, Vt, 0.15), gpu_ctx.extract_tiles(b[0], b[1], U, Vt, 0.15))
    assert gpu_ctx.detect_tiles(a[0], a[1], S, 0.15)[0] == gpu_ctx.detect_tiles(a[0], a[1], S, 0.15)[0]
    f1 = gpu_ctx.ref_embed(host, np.sort(S.reshape(-1))[::-1][:256].copy(), 0.15, 100)
    f2 = gpu_ctx.ref_embed(host, np.sort(S.reshape(-1))[::-1][:256].copy(), 0.15, 100)
    assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])


def test_fallback_tiles_do_not_depend_on_their_wave_partners(gpu_ctx):
    """The fallback kernel takes rank-deficient tiles in the order an atomic counter handed out,
    which varies from run to run and with what else is in the batch.  A tile's result must not
    depend on which tiles share its wave: many flagged tiles (several waves' worth), embedded
    twice, alone and inside a batch with other content -> identical bytes."""
    rng = np.random.default_rng(17)
    H, W = 256, 512                                       # 2048 tiles
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    ty, tx = np.divmod(np.arange(2048), 64)
    for k in range(0, 2048, 3):                           # every third tile degenerate, of varying kinds
        y, x = ty[k] * 8, tx[k] * 8
        kind = k % 4
        if kind == 0: img[y:y + 8, x:x + 8] = rng.integers(0, 256)
        elif kind == 1: img[y:y + 8, x:x + 8] = rng.integers(0, 256, (1, 8))          # rank 1: equal rows
        elif kind == 2: img[y:y + 8, x + 4:x + 8] = img[y:y + 8, x:x + 4]             # repeated columns
        else: img[y:y + 8, x:x + 8] = np.where(np.arange(8)[:, None] < 3, 255, 0)     # saturated edge
    wys = rng.integers(0, 256, (H, W)).astype(np.float32)
    _, S, _ = gpu_ctx.svd_tiles(wys)
    a = gpu_ctx.embed_tiles(img, S, 0.15, want_yw=True)
    b = gpu_ctx.embed_tiles(img, S, 0.15, want_yw=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    batch = np.stack([np.full((H, W), 9, np.uint8), img, rng.integers(0, 256, (H, W), dtype=np.uint8), img[::-1].copy()])
    c = gpu_ctx.embed_tiles(batch, S, 0.15, want_yw=True)
    assert np.array_equal(c[0][1], a[0]) and np.array_equal(c[1][1], a[1]) and np.array_equal(c[2][1], a[2])


def test_structured_tiles_with_repeated_singular_values(gpu_ctx):
    """Checkerboards, diagonals, binary noise, stripes: repeated / zero singular values
    (singular vectors not unique) must still satisfy the defining properties, converge,
    and agree with the oracle on the singular values themselves."""
    H = W = 64
    yy, xx = np.mgrid[0:H, 0:W]
    pats = [((yy + xx) % 2 * 255), ((yy // 2 + xx // 2) % 2 * 255), (np.equal(yy % 8, xx % 8) * 255),
            ((xx % 2) * 255), ((yy % 4 < 2) * 200 + 20), np.random.default_rng(3).integers(0, 2, (H, W)) * 255,
            (np.equal(yy % 8, 7 - xx % 8) * 128 + 64), ((yy % 8) * 32 + (xx % 8) * 3)]
    planes = np.stack(pats).astype(np.uint8)
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    Uo, So, Vto = o.watermark_decompose(wys, 8)
    alpha = 0.15
    stego, sc, yw = gpu_ctx.embed_tiles(planes, So, alpha, want_yw=True)
    assert np.isfinite(yw).all() and np.isfinite(sc).all()
    for p in range(planes.shape[0]):
        X = o.to_tiles(planes[p].astype(np.float64)).reshape(-1, 8, 8)
        sx = np.linalg.svd(X, compute_uv=False)
        scp = sc[p].reshape(-1, 8).astype(np.float64)
        assert np.max((np.abs(scp - sx) - 4 * 2.0 ** -14) / np.maximum(sx[:, :1], 1.0)) < 1e-5, p
        T = o.to_tiles(yw[p].astype(np.float64)).reshape(-1, 8, 8)
        want = np.sort(scp + alpha * So.reshape(-1, 8), axis=1)[:, ::-1]
        got = np.linalg.svd(T, compute_uv=False)
        assert np.max(np.abs(got - want) / np.maximum(want[:, :1], 1.0)) < 2e-4, p
        assert np.array_equal(stego[p], np.clip(yw[p], 0, 255).astype(np.uint8))
    # sigma-only / detect kernels on the same planes
    s = gpu_ctx.sigma_tiles(planes)
    assert np.max(np.abs(s - sc)) < 1e-2
    # clipping at 0 / 255 destroys the mark on saturated patterns (true of the scheme, not of the
    # kernels): check the detect kernel against the oracle's formula on the same stego instead
    got_scores = gpu_ctx.detect_tiles(stego, sc, So, alpha)
    for p in range(planes.shape[0]):
        sh = (o.stego_sigma(stego[p].astype(np.float32), 8) - sc[p]) / alpha
        if np.std(sh) < 1e-2:          # NC of a constant vector is 0/0 (rounding noise on either side)
            assert abs(got_scores[p]) < 1.0 + 1e-9
            continue
        assert abs(got_scores[p] - o.detect_plane(stego[p].astype(np.float32), sc[p], So, alpha, 8)) < 2e-3


def test_two_contexts_from_two_threads(hostapi):
    """one opaque handle per device/stream, independently usable from two
    threads (ctypes releases the GIL during the calls)."""
    import threading
    host, wys = _inputs(256, 384)
    ref = o.embed_plane(host.astype(np.float32), wys, 0.15, 0.6, tile=8)
    results = [None, None]

    def work(i):
        with hostapi.Context(0) as c:
            for _ in range(5):
                st, sc, _ = c.embed_tiles(host, ref["Sw"], 0.15)
                w = c.extract_tiles(st, sc, ref["Uw"], ref["Vwt"], 0.15)
            results[i] = (st, sc, w)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert results[0] is not None and results[1] is not None
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    assert np.abs(results[0][0].astype(int) - ref["stego"].astype(int)).max() <= 1


@pytest.mark.parametrize("K", [8, 5])
def test_extract_with_pixel_domain_factors(gpu_ctx, K):
    """wm_tile_factors_to_pixel_dev + wm_extract_tiles_px_u8_dev: the IDCT is folded into the
    factors once per watermark; the per-frame result matches the DCT-domain entry point and the
    oracle, for several frames sharing the factors and for a ragged plane."""
    alpha = 0.15
    for H, W in ((64, 96), (52, 83)):
        Hb, Wb = H // 8 * 8, W // 8 * 8
        host, wys = _inputs(H, W)
        frames = np.stack([host, np.random.default_rng(3).integers(0, 256, (H, W), dtype=np.uint8)])
        refs = [o.embed_plane(f.astype(np.float32), wys, alpha, kfrac=0.0, tile=8, k_floor=K) for f in frames]
        st = np.stack([r["stego"] for r in refs])
        sc = np.stack([r["Sc"] for r in refs]).reshape(2, -1, 8)
        Uw, Vwt = refs[0]["Uw"].reshape(-1, 8, 8), refs[0]["Vwt"].reshape(-1, 8, 8)
        nt = Uw.shape[0]
        d = {k: gpu_ctx.malloc(v) for k, v in dict(st=st.nbytes, sc=sc.nbytes, U=Uw.nbytes, V=Vwt.nbytes, Ux=Uw.nbytes,
                                                   Vx=Vwt.nbytes, o1=2 * H * W * 4, o2=2 * H * W * 4).items()}
        gpu_ctx.h2d(d["st"], st); gpu_ctx.h2d(d["sc"], np.ascontiguousarray(sc))
        gpu_ctx.h2d(d["U"], np.ascontiguousarray(Uw)); gpu_ctx.h2d(d["V"], np.ascontiguousarray(Vwt))
        gpu_ctx.tile_factors_to_pixel_dev(d["U"], d["V"], d["Ux"], d["Vx"], nt)
        gpu_ctx.extract_tiles_u8_dev(d["st"], d["sc"], d["U"], d["V"], d["o1"], 2, H, W, W, H * W, 0, alpha, K)
        gpu_ctx.extract_tiles_px_u8_dev(d["st"], d["sc"], d["Ux"], d["Vx"], d["o2"], 2, H, W, W, H * W, 0, alpha, K)
        o1 = np.empty((2, H, W), np.float32); o2 = np.empty((2, H, W), np.float32)
        gpu_ctx.d2h(o1, d["o1"]); gpu_ctx.d2h(o2, d["o2"]); gpu_ctx.check_status()
        # the factors themselves: Ux = D^T Uw, Vxt = Vwt D per tile
        Ux = np.empty_like(Uw); Vx = np.empty_like(Vwt)
        gpu_ctx.d2h(Ux, d["Ux"]); gpu_ctx.d2h(Vx, d["Vx"])
        D = o.dct_basis(8).astype(np.float64)
        assert np.abs(Ux - np.einsum("kr,tki->tri", D, Uw)).max() < 2e-6
        assert np.abs(Vx - np.einsum("tik,kc->tic", Vwt, D)).max() < 2e-6
        assert np.abs(o1 - o2).max() < 1e-3 * max(1.0, np.abs(o1).max())
        for f in range(2):
            wo = o.extract_plane(st[f].astype(np.float32), refs[f]["Sc"], refs[0]["Uw"], refs[0]["Vwt"], alpha, 0.0,
                                 H, W, 8, k_floor=K)
            assert np.abs(o2[f] - wo).max() < 2e-2
            assert np.all(o2[f][Hb:, :] == 0) and np.all(o2[f][:, Wb:] == 0)
        # in-place conversion gives the same factors
        gpu_ctx.tile_factors_to_pixel_dev(d["U"], d["V"], d["U"], d["V"], nt)
        U2 = np.empty_like(Uw); gpu_ctx.d2h(U2, d["U"])
        assert np.array_equal(U2, Ux)
        for v in d.values():
            gpu_ctx.free(v)
    with pytest.raises(ValueError):
        gpu_ctx.tile_factors_to_pixel_dev(0, 0, 0, 0, 4)


@pytest.mark.parametrize("noise", [3.0, 1.0, 0.4])
def test_embed_parity_on_smooth_content(gpu_ctx, noise):
    """Camera-like content: a smooth field plus sensor noise, so every tile has a steep singular
    spectrum (s_8 / s_1 down to 1e-4 and below) - the regime where the V-free form divides by the
    smallest singular values.  Same bar as on iid noise: sigma 1e-4 of s_1, stego within 1 LSB."""
    H, W, alpha = 256, 384, 0.15
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:H, 0:W]
    field = 128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 40 * np.sin((xx + 2 * yy) / 91.0)
    host = np.clip(field + rng.normal(0, noise, (H, W)), 0, 255).astype(np.uint8)
    _, wys = _inputs(H, W)
    ref = o.embed_plane(host.astype(np.float32), wys, alpha, kfrac=0.6, tile=8)
    ratio = ref["Sc"][..., 7] / ref["Sc"][..., 0]
    stego, sc, yw = gpu_ctx.embed_tiles(host, ref["Sw"], alpha, want_yw=True)
    assert _rel_sigma(sc, ref["Sc"]) < SIGMA_RTOL
    ok = ratio > 1e-5                     # tiles the fast kernel keeps (below: completion path, arbitrary null vectors)
    assert ok.mean() > 0.5
    mask = np.kron(ok, np.ones((8, 8), bool))
    d = np.abs(stego.astype(int) - ref["stego"].astype(int))
    assert d[mask].max() <= 1
    assert (d[mask] != 0).mean() < 2e-3
    assert np.abs(yw - ref["Yw"])[mask].max() < 0.05


def test_extract_sum_over_planes(gpu_ctx):
    """wm_extract_tiles_sum_u8: frames of a clip are added on the device (ascending order)."""
    H, W, alpha = 64, 96, 0.15
    host, wys = _inputs(H, W)
    frames = np.stack([host, host[::-1].copy(), np.random.default_rng(4).integers(0, 256, (H, W), dtype=np.uint8)])
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    st, sc, _ = gpu_ctx.embed_tiles(frames, S, alpha)
    each = gpu_ctx.extract_tiles(st, sc, U, Vt, alpha)
    tot = gpu_ctx.extract_tiles(st, sc, U, Vt, alpha, sum_planes=True)
    want = (each[0] + each[1]) + each[2]                  # same order, same float32 additions
    assert tot.shape == (H, W) and np.array_equal(tot, want)
    assert np.array_equal(gpu_ctx.extract_tiles(st[:1], sc[:1], U, Vt, alpha, sum_planes=True), each[0])


def test_generated_jacobi_stream_against_the_cpp_form_on_the_gpu(hostapi):
    """The tile kernels run the generated gfx950 instruction stream (csrc/wm_jacobi_gfx950.inc); the same library built
    with -DWM_NO_ASM_JACOBI runs hipcc's code for the C++ form in wm_tile_math.h (what the CPU harness tests).  Same
    inputs through both on the GPU: singular values to float32 rounding, stego within 1 LSB on a handful of pixels,
    sigma-only / extract / detect outputs equal to rounding - noise, smooth, flat and structured content."""
    import ctypes as C
    import __graft_entry__ as ge
    path = ge.build_hip_noasm()
    lib = hostapi.load_library(path)

    class Ctx(hostapi.Context):
        def __init__(self, lib_):
            self.lib = lib_
            h = C.c_void_p()
            assert lib_.wm_create(0, None, C.byref(h)) == 0
            self._h = h; self.device = 0

    rng = np.random.default_rng(99)
    H, W = 256, 512
    yy, xx = np.mgrid[0:H, 0:W]
    planes = np.stack([
        rng.integers(0, 256, (H, W)),
        np.clip(128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 2, (H, W)), 0, 255),
        np.where(xx < W // 2, 200, rng.integers(0, 256, (H, W))),
        (yy + xx) % 2 * 255,
    ]).astype(np.uint8)
    wys = rng.integers(0, 256, (H, W)).astype(np.float32)
    wys[:64] = 77.0                                 # rank-1 tiles and ...
    wys[64:128] = ((yy + xx) % 2 * 255)[64:128]     # ... rank-2 tiles on the watermark side too (the stream with V)
    res, svds, yws = [], [], []
    for ctx in (hostapi.Context(0), Ctx(lib)):
        U, S, Vt = ctx.svd_tiles(wys)
        svds.append((U, S, Vt))
        st, sc, yw_ = ctx.embed_tiles(planes, S, 0.15, want_yw=True)
        yws.append(yw_)
        sig = ctx.sigma_tiles(st)
        w = ctx.extract_tiles(st, sc, U, Vt, 0.15)
        d = ctx.detect_tiles(st, sc, S, 0.15)
        res.append((st, sc, sig, w, d))
        ctx.close()
    (st_a, sc_a, sig_a, w_a, d_a), (st_c, sc_c, sig_c, w_c, d_c) = res
    (U_a, S_a, Vt_a), (U_c, S_c, Vt_c) = svds
    assert np.max(np.abs(S_a - S_c) / np.maximum(S_c[..., :1], 1.0)) < 2e-6
    rec = lambda U_, S_, Vt_: np.einsum("...ri,...i,...ic->...rc", U_, S_, Vt_)
    assert np.max(np.abs(rec(U_a, S_a, Vt_a) - rec(U_c, S_c, Vt_c))) < 2e-3     # DCT coefficients up to 2040: 1e-6 relative
    assert np.max(np.abs(np.einsum("...ic,...jc->...ij", Vt_a, Vt_a) - np.eye(8))) < 1e-5
    assert np.max(np.abs(sc_a - sc_c) / np.maximum(sc_c[..., :1], 1.0)) < 2e-6
    dd = np.abs(st_a.astype(int) - st_c.astype(int))
    assert dd[:3].max() <= 1 and np.mean(dd[:3] != 0) < 1e-4
    # the checkerboard's tiles have a REPEATED singular value (255 * 4 twice): the basis inside that plane is arbitrary,
    # the two builds may pick different ones and the injected alpha * (sw_1 u_1 v_1^T + sw_2 u_2 v_2^T) differs with it.
    # What is defined there is the reference's invariant, for each build on its own:
    nby, nbx = H // 8, W // 8
    for yw_, sc_, S_ in zip(yws, (sc_a, sc_c), (S_a, S_c)):
        Tt = yw_[3].reshape(nby, 8, nbx, 8).transpose(0, 2, 1, 3).reshape(-1, 8, 8).astype(np.float64)
        got = np.linalg.svd(Tt, compute_uv=False)
        want = np.sort(sc_[3].reshape(-1, 8).astype(np.float64) + 0.15 * S_.reshape(-1, 8), axis=1)[:, ::-1]
        assert np.max(np.abs(got - want) / np.maximum(want[:, :1], 1.0)) < 1e-4
    # sigma-only kernels on IDENTICAL input (the asm library's stego through both)
    ctx = Ctx(lib)
    sig_c2 = ctx.sigma_tiles(st_a)
    ctx.close()
    assert np.max(np.abs(sig_a - sig_c2) / np.maximum(sig_c2[..., :1], 1.0)) < 2e-5    # skip threshold 1e-8: bounded at 5e-5 s_i
    assert np.abs(d_a - d_c)[:3].max() < 2e-3          # (plane 3: two different valid stegos, see above)


def test_device_dct_against_the_published_jpeg_example(gpu_ctx):
    """The device's DCT on an external known answer: the watermark-side kernel (K3) factors dct2(tile) = U diag(S) Vt,
    so its product must reproduce the JPEG literature's printed coefficients of the worked example
    (tests/golden/external/jpeg_dct_example.npz) to the two decimals they are printed with."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "external", "jpeg_dct_example.npz"))
    plane = np.tile(g["block"].astype(np.float32), (2, 3))                 # six copies: a 16 x 24 plane
    U, S, Vt = gpu_ctx.svd_tiles(plane)
    for ty in range(2):
        for tx in range(3):
            C = (U[ty, tx] * S[ty, tx]) @ Vt[ty, tx]
            assert np.abs(C - g["dct"]).max() < 6e-3


def test_tile_singular_values_against_a_published_example(gpu_ctx):
    """External known answer for the tile SVD on the device: the 4 x 5 example of Wikipedia's 'Singular value
    decomposition' article (singular values 3, sqrt 5, 2, 0) zero-padded to an 8 x 8 uint8 tile, scaled by 1, 17 and 63
    (so that the values are exact multiples), through the sigma-only kernel and through embed's Sc side output
    (a rank-3 tile: the fallback path)."""
    from test_oracle import _wikipedia_svd_tile
    m, want = _wikipedia_svd_tile()
    plane = np.zeros((8, 64), np.uint8)
    scales = [1, 17, 63, 1, 17, 63, 1, 17]
    for k, sc_ in enumerate(scales):
        plane[:, 8 * k:8 * k + 8] = m * sc_
    s = gpu_ctx.sigma_tiles(plane)[0]
    for k, sc_ in enumerate(scales):
        assert np.abs(s[k] - want * sc_).max() < 2e-5 * 3 * sc_ + 1e-5, k
    _, sc, _ = gpu_ctx.embed_tiles(plane, np.zeros((1, 8, 8), np.float32), 0.15)
    for k, sc_ in enumerate(scales):
        assert np.abs(sc[0, k] - want * sc_).max() < 2e-4 * 3 * sc_ + 1e-3, k        # completion pattern: 2^-14 added to the DCT tile


def test_bad_args_of_the_round2_entry_points(gpu_ctx, hostapi):
    """NULL / out-of-range arguments of the entry points added in round 2 come back as WM_ERR_BADARG (ValueError),
    never as a launch: raw C ABI."""
    import ctypes as C
    lib, h = gpu_ctx.lib, gpu_ctx._h
    vp = C.c_void_p
    d = gpu_ctx.malloc(4096)
    try:
        for name, args in (
            ("wm_permute_u8_f32_dev", (None, vp(d), vp(d), 16, 1)),
            ("wm_permute_f32_dev", (vp(d), vp(d), vp(d), 16, 1)),                      # in place
            ("wm_unpermute_f32_dev", (vp(d), None, vp(d + 1024), 16, 1)),
            ("wm_permute_f32_dev", (vp(d), vp(d + 2048), vp(d + 1024), 1 << 31, 1)),    # index is int32
            ("wm_unpermute_f32_dev", (vp(d), vp(d + 2048), vp(d + 1024), 16, 70000)),
            ("wm_ref_sigma_planes_u8_dev", (None, vp(d), 1, 8, 8, 8, 64)),
            ("wm_ref_sigma_planes_u8_dev", (vp(d), vp(d + 1024), 0, 8, 8, 8, 64)),
            ("wm_ref_embed_planes_u8_dev", (vp(d), vp(d + 1024), vp(d + 2048), None, None, 1, 8, 8, 8, 64, 0, 0.1, 4)),
            ("wm_ref_embed_planes_u8_dev", (vp(d), vp(d + 1024), vp(d + 2048), vp(d + 3072), None, 1, 8, 8, 8, 64, 0, 0.1, 9)),
            ("wm_ref_extract_planes_u8_dev", (vp(d), vp(d + 1024), None, vp(d + 2048), vp(d + 3072), 1, 8, 8, 8, 64, 0.1, 4)),
            ("wm_ref_detect_planes_u8_dev", (vp(d), vp(d + 1024), vp(d + 2048), None, 1, 8, 8, 8, 64, 0.1)),
            ("wm_ref_sigma_planes_u8_dev", (vp(d), vp(d + 1024), 1, 8, 8, 4, 64)),      # row_stride < W
        ):
            rc = getattr(lib, name)(h, *args)
            assert rc == hostapi.WM_ERR_BADARG, (name, rc, lib.wm_last_error())
        # zero-sized permute is a no-op
        assert lib.wm_permute_f32_dev(h, None, None, None, 0, 1) == hostapi.WM_OK
    finally:
        gpu_ctx.free(d)
    gpu_ctx.check_status()


def test_random_geometries_against_the_oracle(gpu_ctx):
    """A sweep of odd geometries through the C ABI: plane counts that do not divide the XCD mapping, tile counts that do
    not fill a wave, strided views, shared and per-plane watermark values, K from 0 to 8 - embed, sigma, extract, detect
    against the oracle on every plane (sizes the oracle does in milliseconds)."""
    rng = np.random.default_rng(2024)
    for case in range(14):
        n = int(rng.integers(1, 6)); H = int(rng.integers(8, 150)); W = int(rng.integers(8, 210))
        K = int(rng.integers(0, 9)); alpha = float(rng.uniform(0.05, 0.3))
        pad_r, pad_c = int(rng.integers(0, 9)), int(rng.integers(0, 13))
        big = rng.integers(0, 256, (n, H + pad_r, W + pad_c), dtype=np.uint8)
        hosts = big[:, pad_r // 2:pad_r // 2 + H, pad_c // 2:pad_c // 2 + W]          # a strided view unless both pads are 0
        wys = rng.integers(0, 256, (H, W)).astype(np.float32)
        Uo, So, Vto = o.watermark_decompose(wys, 8)
        per_plane = bool(case % 3 == 0)
        Sw = np.stack([So * (1.0 + 0.1 * p) for p in range(n)]).astype(np.float32) if per_plane else So
        stego, sc, yw = gpu_ctx.embed_tiles(hosts, Sw, alpha, K=K, want_yw=True)
        sig = gpu_ctx.sigma_tiles(stego)
        ext = gpu_ctx.extract_tiles(stego, sc, Uo, Vto, alpha, K=K)
        det = gpu_ctx.detect_tiles(stego, sc, Sw, alpha)
        Hb, Wb = H // 8 * 8, W // 8 * 8
        for p in range(n):
            swp = Sw[p] if per_plane else So
            ref = o.embed_plane(hosts[p].astype(np.float32), wys, alpha, 0.0, 8, k_floor=K, wm_svd=(Uo, swp, Vto))
            assert _rel_sigma(sc[p], ref["Sc"]) < SIGMA_RTOL, (case, p)
            d = np.abs(stego[p].astype(int) - ref["stego"].astype(int))
            assert d.max() <= 1, (case, p, n, H, W, K)                 # (K = 0: nothing is injected, both reproduce the host)
            assert np.array_equal(stego[p][Hb:], hosts[p][Hb:]) and np.array_equal(stego[p][:, Wb:], hosts[p][:, Wb:])
            if Hb and Wb:
                assert _rel_sigma(sig[p], o.stego_sigma(stego[p].astype(np.float32), 8)) < SIGMA_RTOL
                wo = o.extract_plane(stego[p].astype(np.float32), sc[p], Uo, Vto, alpha, 0.0, H, W, 8, k_floor=K)
                assert np.abs(ext[p] - wo).max() < 5e-2 * max(1.0, 0.15 / alpha), (case, p)
                if K > 0:     # with nothing injected the NC is a correlation of rounding residues: not comparable
                    assert abs(det[p] - o.detect_plane(stego[p].astype(np.float32), sc[p], swp, alpha, 8)) < 1e-4
