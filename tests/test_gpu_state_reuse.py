"""A long-lived context must behave like a fresh one: its grow-only scratch, workspace carving,
cached DCT bases, fallback list and status words are reused across calls of different shapes.
Random sequences of operations on ONE context are compared, call by call and bit for bit, with
the same call on a brand-new context."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8), (16, 40), (64, 96), (96, 64), (40, 40), (128, 72), (52, 83), (72, 128), (33, 17), (5, 20), (9, 7), (1, 1)]


def _same(a, b):
    if isinstance(a, tuple):
        return all(_same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is b
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and np.array_equal(a, b)
    return a == b


def test_random_call_sequences_match_fresh_contexts(hostapi):
    rng = np.random.default_rng(2024)
    long_lived = hostapi.Context(0)
    try:
        for step in range(60):
            H, W = SHAPES[rng.integers(len(SHAPES))]
            n = int(rng.integers(1, 4))
            planes = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
            if rng.random() < 0.3:                       # rank-deficient content -> fallback list in use
                planes[:, : H // 2] = planes[:, :1, :1]
            wys = rng.integers(0, 256, (H, W)).astype(np.float32)
            op = int(rng.integers(7))
            L = min(H, W)

            def run(ctx):
                if op == 0:
                    _, S, _ = ctx.svd_tiles(wys)
                    return ctx.embed_tiles(planes, S, 0.15, int(rng_k), want_yw=True)
                if op == 1:
                    return ctx.sigma_tiles(planes)
                if op == 2:
                    U, S, Vt = ctx.svd_tiles(wys)
                    st, sc, _ = ctx.embed_tiles(planes, S, 0.15, 8)
                    return ctx.extract_tiles(st, sc, U, Vt, 0.15, 8), ctx.detect_tiles(st, sc, S, 0.15)
                if op == 3:
                    U, S, Vt = ctx.ref_svd(wys, apply_dct=True)
                    st, sc, yw = ctx.ref_embed_planes(planes, S, 0.15, max(1, L // 2), want_yw=True)
                    return st, sc, yw, ctx.ref_extract_planes(st, sc, U, Vt, 0.15, max(1, L // 2))
                if op == 4:
                    return ctx.ref_sigma_planes(planes), ctx.ref_sigma(planes[0])
                if op == 5:
                    img = np.ascontiguousarray(np.moveaxis(np.concatenate([planes, planes, planes])[:3], 0, -1))
                    ycc = ctx.color("bgr2ycrcb", img)
                    return ycc, ctx.color("ycrcb2bgr", ycc), ctx.color("bgr2gray", img), ctx.psnr(img, ycc)
                f = wys - 100.0
                return ctx.normalize_u8(f, True), ctx.normalize_u8(f, False), ctx.ssim(planes[0], wys)

            rng_k = rng.integers(1, 9)
            got = run(long_lived)
            with hostapi.Context(0) as fresh:
                want = run(fresh)
            assert _same(got, want), f"step {step}: op {op} on {n}x{H}x{W} differs from a fresh context"
    finally:
        long_lived.close()


def test_identity_cases(gpu_ctx):
    """alpha = 0 or K = 0 must return the host bytes unchanged, in both modes, also on ragged and
    tiny planes (H or W below one tile: tile mode has nothing to do, full-frame still runs)."""
    rng = np.random.default_rng(3)
    for H, W in ((64, 96), (52, 83), (5, 20), (9, 7), (1, 1)):
        x = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
        sw_t = np.abs(rng.normal(0, 100, (H // 8, W // 8, 8))).astype(np.float32)
        for alpha, K in ((0.0, 8), (0.15, 0)):
            st, sc, _ = gpu_ctx.embed_tiles(x, sw_t, alpha, K)
            assert np.array_equal(st, x)
        L = min(H, W)
        sw_f = np.sort(np.abs(rng.normal(0, 1000, L)).astype(np.float32))[::-1].copy()
        for alpha, K in ((0.0, L), (0.15, 0)):
            st, sc, _ = gpu_ctx.ref_embed_planes(x, sw_f, alpha, K)
            assert np.array_equal(st, x)
            ref = np.linalg.svd(x[0].astype(np.float64), compute_uv=False)
            assert np.abs(sc[0] - ref).max() <= 2e-6 * max(ref[0], 1.0)
