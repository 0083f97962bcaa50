"""Quick device-resident timing of the tile kernels (no torch): HIP events on the
context's stream around repeated launches.  Development aid; bench.py is the
contract benchmark."""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module(
    "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=2160)
    ap.add_argument("--W", type=int, default=3840)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=0.15)
    ap.add_argument("--content", default="noise", choices=["noise", "natural", "screen", "flat", "letterbox"],
                    help="noise: iid uint8; natural: smooth field + sensor noise; screen: flat rectangles "
                         "+ 1-px strokes (mostly rank-deficient tiles); flat: one constant; letterbox: natural content "
                         "between black bars (2.39:1 in 16:9: a quarter of the tiles constant)")
    a = ap.parse_args()
    H, W, F = a.H, a.W, a.frames
    nt = (H // 8) * (W // 8)
    ctx = api.Context(0)
    rng = np.random.default_rng(1234)
    if a.content == "noise":
        host = rng.integers(0, 256, (F, H, W), dtype=np.uint8)
    elif a.content in ("natural", "letterbox"):
        low = rng.uniform(20, 235, (F, H // 16 + 2, W // 16 + 2)).astype(np.float32)
        up = np.kron(low, np.ones((16, 16), np.float32))[:, 8:8 + H, 8:8 + W]
        for ax in (1, 2):                       # box blur 16 -> piecewise-linear field
            c = np.cumsum(up, axis=ax)
            up = (np.take(c, np.arange(16, c.shape[ax]), axis=ax) - np.take(c, np.arange(0, c.shape[ax] - 16), axis=ax)) / 16
            pad = [(0, 0)] * 3; pad[ax] = (8, 8); up = np.pad(up, pad, mode="edge")
        host = np.clip(up + rng.normal(0, 2.0, up.shape), 0, 255).astype(np.uint8)
        if a.content == "letterbox":
            bar = int(round((H - W / 2.39) / 2 / 8)) * 8
            host[:, :bar] = 16; host[:, H - bar:] = 16
    elif a.content == "screen":
        host = np.full((F, H, W), 240, np.uint8)
        for f in range(F):
            for _ in range(200):
                y, x = rng.integers(0, H - 64), rng.integers(0, W - 64)
                h, w = rng.integers(8, 400), rng.integers(8, 400)
                host[f, y:y + h, x:x + w] = rng.integers(0, 256)
            host[f, ::37, :] = 0; host[f, :, ::53] = 0
    else:
        host = np.full((F, H, W), 128, np.uint8)
    host = np.ascontiguousarray(host)
    nz = [int(np.linalg.matrix_rank(host[0, y:y + 8, x:x + 8].astype(np.float64))) for y in range(0, 512, 8) for x in range(0, 512, 8)]
    print(f"content={a.content}: mean tile rank (512x512 corner) {np.mean(nz):.2f}, full-rank fraction {np.mean(np.array(nz) == 8):.3f}")
    wys = rng.integers(0, 256, (H, W)).astype(np.float32)
    d_host = ctx.malloc(host.nbytes); ctx.h2d(d_host, host)
    d_stego = ctx.malloc(host.nbytes)
    d_wys = ctx.malloc(wys.nbytes); ctx.h2d(d_wys, wys)
    d_U = ctx.malloc(nt * 64 * 4); d_V = ctx.malloc(nt * 64 * 4); d_S = ctx.malloc(nt * 8 * 4)
    d_sc = ctx.malloc(F * nt * 8 * 4)
    d_out = ctx.malloc(F * H * W * 4)
    d_scores = ctx.malloc(F * 8)

    def timed(name, fn, bytes_per_call):
        fn(); ctx.sync()
        ctx.event_record(0)
        for _ in range(a.reps):
            fn()
        ctx.event_record(1)
        ms = ctx.event_elapsed_ms(0, 1) / a.reps
        print(f"{name:10s} {ms*1e3:10.1f} us/launch  {ms*1e3/F:9.1f} us/frame  "
              f"{F/ms*1e3:10.0f} frames/s  {bytes_per_call/ms/1e9*1e3:8.1f} GB/s algorithmic", flush=True)
        return ms

    P = H * W
    timed("svd_wm", lambda: ctx.svd_tiles_f32_dev(d_wys, d_U, d_S, d_V, 1, H, W, W, H * W), 12.5 * P / F * F)
    timed("embed", lambda: ctx.embed_tiles_u8_dev(d_host, d_S, d_stego, d_sc, None, F, H, W, W, H * W, 0, a.alpha, 8), 3.0 * P * F)
    timed("sigma", lambda: ctx.sigma_tiles_u8_dev(d_stego, d_sc, F, H, W, W, H * W), 1.5 * P * F)
    ctx.embed_tiles_u8_dev(d_host, d_S, d_stego, d_sc, None, F, H, W, W, H * W, 0, a.alpha, 8)
    timed("extract", lambda: ctx.extract_tiles_u8_dev(d_stego, d_sc, d_U, d_V, d_out, F, H, W, W, H * W, 0, a.alpha, 8), 10.5 * P * F)
    d_Ux = ctx.malloc(nt * 64 * 4); d_Vx = ctx.malloc(nt * 64 * 4)
    timed("factors_px", lambda: ctx.tile_factors_to_pixel_dev(d_U, d_V, d_Ux, d_Vx, nt), 4.0 * nt * 256)
    timed("extract_px", lambda: ctx.extract_tiles_px_u8_dev(d_stego, d_sc, d_Ux, d_Vx, d_out, F, H, W, W, H * W, 0, a.alpha, 8), 10.5 * P * F)
    timed("detect", lambda: ctx.detect_tiles_u8_dev(d_stego, d_sc, d_S, d_scores, F, H, W, W, H * W, 0, a.alpha), 2.0 * P * F)
    # pixel-side streaming kernels (device-resident): algorithmic bytes / HIP-event time
    n_px = F * H * W
    d_bgr = ctx.malloc(n_px * 3); d_bgr2 = ctx.malloc(n_px * 3); d_y = ctx.malloc(n_px)
    ctx.memset(d_bgr, 0x5a, n_px * 3)
    d_ssd = ctx.malloc(64); d_ss = ctx.malloc(64)
    timed("bgr2ycc", lambda: ctx._call("wm_bgr_to_ycrcb_u8_dev", d_bgr, d_bgr2, n_px), 6.0 * n_px)
    timed("bgr2y", lambda: ctx._call("wm_bgr_to_y_u8_dev", d_bgr, d_y, n_px), 4.0 * n_px)
    timed("replace_y", lambda: ctx._call("wm_replace_y_u8_dev", d_bgr, d_y, d_bgr2, n_px), 7.0 * n_px)
    timed("sqdiff", lambda: ctx._call("wm_sqdiff_u8_dev", d_bgr, d_bgr2, n_px * 3, d_ssd), 6.0 * n_px)
    timed("ssim", lambda: [ctx._call("wm_ssim_dev", d_host + i * H * W, W, d_stego + i * H * W, W, H, W, 0, d_ss)
                           for i in range(F)], 2.0 * n_px)
    timed("norm_u8", lambda: ctx._call("wm_normalize_u8_dev", d_out, n_px, 1, d_y), 9.0 * n_px)
    ctx.check_status()
    sc = np.zeros(F, np.float64); ctx.d2h(sc, d_scores)
    print("detect scores", sc[:4])
    ctx.close()


if __name__ == "__main__":
    main()
