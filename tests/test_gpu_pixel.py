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
