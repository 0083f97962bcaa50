"""Pixel-side kernels vs the oracle: colour conversion bit-exact (integer fixed
point), PSNR / SSIM to float tolerance, min-max normalise to 1 LSB."""
import numpy as np
import pytest

from oracle import wm_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (37, 53), (256, 320), (1080, 1920)])
def test_colour_conversions_bit_exact(gpu_ctx, shape):
    H, W = shape
    img = np.random.default_rng(H * 1000 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    if H * W > 64:      # saturation corners
        img[0, 0] = (255, 255, 255); img[0, 1] = (0, 0, 0); img[0, 2] = (255, 0, 0); img[0, 3] = (0, 0, 255)
        img[0, 4] = (0, 255, 0)
    ycc = o.bgr_to_ycrcb(img)
    assert np.array_equal(gpu_ctx.color("bgr2ycrcb", img), ycc)
    assert np.array_equal(gpu_ctx.color("ycrcb2bgr", img), o.ycrcb_to_bgr(img))
    assert np.array_equal(gpu_ctx.color("bgr2gray", img), o.bgr_to_gray(img))
    assert np.array_equal(gpu_ctx.color("bgr2y", img), ycc[..., 0])
    ynew = np.random.default_rng(1).integers(0, 256, (H, W), dtype=np.uint8)
    want = ycc.copy(); want[..., 0] = ynew
    assert np.array_equal(gpu_ctx.color("replace_y", img, ynew), o.ycrcb_to_bgr(want))


def test_psnr_ssim_normalize(gpu_ctx):
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (300, 421, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8)
    assert abs(gpu_ctx.psnr(a, b) - o.psnr(a, b)) < 1e-4
    assert gpu_ctx.psnr(a, a) == 99.0
    ga, gb = o.bgr_to_gray(a), o.bgr_to_gray(b)
    assert abs(gpu_ctx.ssim(ga, gb) - o.ssim(ga, gb)) < 2e-5
    yw = gb.astype(np.float32) + rng.normal(0, 0.3, gb.shape).astype(np.float32)   # gray-mode SSIM: (u8, float Yw)
    assert abs(gpu_ctx.ssim(ga, yw) - o.ssim(ga, yw)) < 2e-5
    assert abs(gpu_ctx.ssim(ga, ga) - 1.0) < 1e-6
    small = rng.integers(0, 256, (7, 9), dtype=np.uint8)                               # smaller than the 11-tap window
    assert abs(gpu_ctx.ssim(small, small[::-1].copy()) - o.ssim(small, small[::-1].copy())) < 2e-5
    x = rng.normal(40, 90, (211, 173)).astype(np.float32)
    want = np.clip(o.normalize_minmax(x), 0, 255).astype(np.uint8)
    got = gpu_ctx.normalize_u8(x, True)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and np.mean(got != want) < 1e-3
    assert np.array_equal(gpu_ctx.normalize_u8(x, False), np.clip(x, 0, 255).astype(np.uint8))
    assert np.array_equal(gpu_ctx.normalize_u8(np.full((4, 4), 3.0, np.float32), True), np.zeros((4, 4), np.uint8))


def test_device_scramble_and_unscramble_are_bit_identical_to_the_host_glue(gpu_ctx):
    """single:66-80 on the device: the permutation stays NumPy's PCG64 shuffle (host, bit-exact by construction),
    the two index passes - flat[idx] and the inverse scatter - and the normalise that follows run as kernels.
    Bit for bit the host glue's result, for uint8 and float32 planes, one plane and three."""
    import importlib
    from conftest import PKG_NAME
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    rng = np.random.default_rng(17)
    for (H, W) in ((40, 56), (1080, 1920)):
        key = hg.derive_key("pw", bytes(range(8)))
        idx = hg.permutation_index(H, W, key)
        g8 = rng.integers(0, 256, (H, W), dtype=np.uint8)
        assert np.array_equal(gpu_ctx.permute_planes(g8, idx), hg.permute(g8.astype(np.float32), idx))
        f3 = rng.uniform(-50, 300, (3, H, W)).astype(np.float32)
        p3 = gpu_ctx.permute_planes(f3, idx)
        for z in range(3):
            assert np.array_equal(p3[z], hg.permute(f3[z], idx))
        for norm in (True, False):
            u3 = gpu_ctx.unpermute_normalize_u8(f3, idx, norm)
            for z in range(3):
                want = gpu_ctx.normalize_u8(hg.unpermute(f3[z], idx), norm)     # the round-1 path: host scatter, device normalise
                assert np.array_equal(u3[z], want)
        assert np.array_equal(gpu_ctx.unpermute_normalize_u8(p3[0], idx, False),
                              np.clip(f3[0], 0, 255).astype(np.uint8))          # unpermute(permute(x)) == x
    # a second key on the same context: the device index cache must not hand out the first key's index
    idx2 = hg.permutation_index(40, 56, hg.derive_key("other", bytes(8)))
    g8 = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    assert np.array_equal(gpu_ctx.permute_planes(g8, idx2), hg.permute(g8.astype(np.float32), idx2))
    with pytest.raises(ValueError):
        gpu_ctx.permute_planes(g8, idx2[:-1])


def test_routed_unscramble_normalise_is_bit_identical_to_the_index_pass(gpu_ctx):
    """wm_route (csrc/wm_route.hip): the random permutation factored once per key into two block-local permutations
    around a block transpose, so that `_unpermute` + normalise + uint8 (single:74-80, 221-222) stream.  Bytes must
    equal the literal chain - host `inv[idx] = arange; flat[inv]`, then the device normalise - for plane sizes below,
    at and above the route's 32768-element block, odd element counts, one plane and several (per-plane min / max),
    with and without normalisation, and straight through the C ABI against wm_unpermute_f32_dev + wm_normalize_u8_dev."""
    import importlib
    from conftest import PKG_NAME
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    hostapi = importlib.import_module(PKG_NAME + ".hostapi")
    vp = hostapi._vp
    rng = np.random.default_rng(23)
    c = gpu_ctx
    for (H, W, n_pl) in ((8, 8, 1), (40, 56, 2), (181, 181, 3), (128, 256, 1), (128, 257, 2), (200, 328, 3), (1080, 1920, 2), (2160, 3840, 1)):
        n = H * W
        idx = hg.permutation_index(H, W, hg.derive_key(f"route{H}x{W}", bytes(8)))
        assert c.route_dev(idx) is not None
        x = (rng.normal(20, 80, (n_pl, H, W)) * rng.uniform(0.2, 3, (n_pl, 1, 1))).astype(np.float32)
        for norm in (True, False):
            got = c.unpermute_normalize_u8(x, idx, norm)
            for z in range(n_pl):
                want = c.normalize_u8(hg.unpermute(x[z], idx), norm)
                assert np.array_equal(got[z], want), (H, W, z, norm)
        g8 = rng.integers(0, 256, (n_pl, H, W), dtype=np.uint8)            # the scramble direction through the same route
        ps = c.permute_planes(g8, idx)
        for z in range(n_pl):
            assert np.array_equal(ps[z], hg.permute(g8[z].astype(np.float32), idx)), (H, W, z)
        # the two device chains side by side through the C ABI (no host scatter involved)
        d_x = c.malloc(x.nbytes); c.h2d(d_x, x)
        n_pad = (n + 3) & ~3
        d_t = c.malloc(n_pad * 4); d_a = c.malloc(n_pl * n); d_b = c.malloc(n_pad)
        c._call("wm_unpermute_normalize_u8_dev", vp(d_x), vp(c.route_dev(idx)), vp(d_a), n, n_pl, 1)
        a = np.empty((n_pl, n), np.uint8); c.d2h(a, d_a)
        for z in range(n_pl):
            c._call("wm_unpermute_f32_dev", vp(d_x + z * n * 4), vp(c.index_dev(idx)), vp(d_t), n, 1)
            c._call("wm_normalize_u8_dev", vp(d_t), n, 1, vp(d_b))
            b = np.empty(n, np.uint8); c.d2h(b, d_b)
            assert np.array_equal(a[z], b), (H, W, z)
        for d in (d_x, d_t, d_a, d_b):
            c.free(d)
    # constant plane: range 0 -> all zeros like cv2.normalize; a route refuses what is not a permutation
    idx = hg.permutation_index(40, 56, hg.derive_key("k", bytes(8)))
    assert not c.unpermute_normalize_u8(np.full((40, 56), 7.5, np.float32), idx, True).any()
    bad = np.arange(40 * 56, dtype=np.int32); bad[5] = bad[6]
    d_bad = c.malloc(bad.nbytes); c.h2d(d_bad, bad)
    r = vp()
    with pytest.raises(ValueError):
        c._call("wm_route_create_dev", vp(d_bad), bad.size, __import__("ctypes").byref(r))
    bad[5] = 40 * 56 + 3; c.h2d(d_bad, bad)
    with pytest.raises(ValueError):
        c._call("wm_route_create_dev", vp(d_bad), bad.size, __import__("ctypes").byref(r))
    c.free(d_bad)


def test_one_call_extract_equals_the_three_step_chain(gpu_ctx):
    """wm_extract_unscrambled_u8_dev (single:203-222 per plane in one call; the extract kernel leaves its own min / max,
    no separate min-max pass) against wm_extract_tiles[_px]_u8_dev + wm_unpermute_normalize_u8_dev: identical bytes -
    full and ragged plane sizes (zeros outside the tile grid take part in the min / max), one plane and several,
    DCT-domain and pixel-domain factors, with and without normalisation."""
    import importlib
    from conftest import PKG_NAME
    hg = importlib.import_module(PKG_NAME + ".hostglue")
    hostapi = importlib.import_module(PKG_NAME + ".hostapi")
    vp = hostapi._vp
    rng = np.random.default_rng(29)
    c = gpu_ctx
    for (H, W, n_pl) in ((64, 96, 1), (128, 520, 3), (70, 101, 2), (1080, 1920, 2)):
        n = H * W; nby, nbx = H // 8, W // 8; nt = nby * nbx
        idx = hg.permutation_index(H, W, hg.derive_key(f"one{H}x{W}", bytes(8)))
        route = c.route_dev(idx)
        stego = rng.integers(0, 256, (n_pl, H, W), dtype=np.uint8)
        wys = rng.integers(0, 256, (H, W)).astype(np.float32)
        U, S, Vt = c.svd_tiles(wys)
        sc = (c.sigma_tiles(stego) * rng.uniform(0.9, 1.0, (n_pl, nby, nbx, 8))).astype(np.float32)
        d_st = c.malloc(stego.nbytes); c.h2d(d_st, stego)
        d_sc = c.malloc(sc.nbytes); c.h2d(d_sc, sc)
        d_u = c.malloc(U.nbytes); c.h2d(d_u, U); d_v = c.malloc(Vt.nbytes); c.h2d(d_v, Vt)
        d_ux = c.malloc(U.nbytes); d_vx = c.malloc(Vt.nbytes)
        c.tile_factors_to_pixel_dev(d_u, d_v, d_ux, d_vx, nt)
        d_w = c.malloc(n_pl * n * 4); d_a = c.malloc(n_pl * n); d_b = c.malloc(n_pl * n)
        for px in (0, 1):
            fu, fv = (d_ux, d_vx) if px else (d_u, d_v)
            for norm in (1, 0):
                if px:
                    c.extract_tiles_px_u8_dev(d_st, d_sc, fu, fv, d_w, n_pl, H, W, W, n, 0, 0.15, 8)
                else:
                    c.extract_tiles_u8_dev(d_st, d_sc, fu, fv, d_w, n_pl, H, W, W, n, 0, 0.15, 8)
                c._call("wm_unpermute_normalize_u8_dev", vp(d_w), vp(route), vp(d_a), n, n_pl, norm)
                a = np.empty((n_pl, n), np.uint8); c.d2h(a, d_a)
                c._call("wm_extract_unscrambled_u8_dev", vp(d_st), vp(d_sc), vp(fu), vp(fv), vp(route), vp(d_b), n_pl, H, W, W, n, 0,
                        0.15, 8, px, norm)
                b = np.empty((n_pl, n), np.uint8); c.d2h(b, d_b)
                assert np.array_equal(a, b), (H, W, n_pl, px, norm)
                assert norm == 0 or (a.min(axis=1) == 0).all() and (a.max(axis=1) >= 254).all()
        c.check_status()
        for d in (d_st, d_sc, d_u, d_v, d_ux, d_vx, d_w, d_a, d_b):
            c.free(d)


def test_ssim_geometries_and_dtypes_against_the_oracle(gpu_ctx):
    """k_ssim walks 64-column strips in bands of 34 rows with an 11-row register ring, reflect-101 borders, buffer-resource
    addressing and a per-plane partial sum: sizes on, one short of and one past the strip / band / ring boundaries, planes
    narrower than the halo, every uint8 / float32 combination of single:44-57's two arguments."""
    rng = np.random.default_rng(23)
    sizes = [(34, 64), (35, 65), (33, 63), (68, 128), (69, 129), (11, 11), (12, 10), (10, 12), (1, 40), (40, 1), (2, 2),
             (45, 200), (101, 75), (136, 257)]
    for (H, W) in sizes:
        a = rng.integers(0, 256, (H, W), dtype=np.uint8)
        b = np.clip(a.astype(int) + rng.integers(-25, 26, a.shape), 0, 255).astype(np.uint8)
        af = a.astype(np.float32) + rng.normal(0, 0.4, a.shape).astype(np.float32)
        bf = b.astype(np.float32) + rng.normal(0, 0.4, b.shape).astype(np.float32)
        for x, y in ((a, b), (a, bf), (af, b), (af, bf)):
            got, want = gpu_ctx.ssim(x, y), o.ssim(x, y)
            assert abs(got - want) < 3e-5, (H, W, x.dtype, y.dtype, got, want)
    # the device entry point with a row stride larger than the width (a window of a larger plane)
    H, W, S = 70, 100, 160
    big1 = rng.integers(0, 256, (H, S), dtype=np.uint8); big2 = rng.integers(0, 256, (H, S), dtype=np.uint8)
    d1 = gpu_ctx.malloc(big1.nbytes); d2 = gpu_ctx.malloc(big2.nbytes); ds = gpu_ctx.malloc(8)
    try:
        gpu_ctx.h2d(d1, big1); gpu_ctx.h2d(d2, big2)
        from conftest import PKG_NAME
        api = __import__("importlib").import_module(PKG_NAME + ".hostapi")
        gpu_ctx._call("wm_ssim_dev", api._vp(d1), S, api._vp(d2), S, H, W, 0, api._vp(ds))
        v = np.zeros(1); gpu_ctx.d2h(v, ds)
        assert abs(float(v[0]) - o.ssim(big1[:, :W].copy(), big2[:, :W].copy())) < 3e-5
        with pytest.raises(ValueError):
            gpu_ctx._call("wm_ssim_dev", api._vp(d1), W - 1, api._vp(d2), S, H, W, 0, api._vp(ds))      # row stride < W
    finally:
        gpu_ctx.free(d1); gpu_ctx.free(d2); gpu_ctx.free(ds)
