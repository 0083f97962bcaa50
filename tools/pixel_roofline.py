"""Roofline of the HBM-bound kernels either side of the hot path (csrc/wm_pixel.hip): keyed scramble / unscramble
(single:66-80), min-max normalise (single:221-222), colour conversions (single:21-30), PSNR (single:38-42), SSIM
(single:44-57).  Device-resident, HIP events on the context's stream; algorithmic bytes / time / 8 TB/s.
Run bare for the table, or under rocprofv3 (`--kernel-trace --stats`, then `--pmc FETCH_SIZE WRITE_SIZE` in its own pass)
for the per-kernel durations and HBM traffic that profiles/r03_pixel_* hold.
    python tools/pixel_roofline.py [--H 2160 --W 3840 --frames 8 --reps 10] [--json out.json]"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd"
api = importlib.import_module(PKG + ".hostapi")
hg = importlib.import_module(PKG + ".hostglue")
HBM = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    H, W, F = a.H, a.W, a.frames
    n = H * W
    ctx = api.Context(0)
    rng = np.random.default_rng(7)
    idx = hg.permutation_index(H, W, hg.derive_key("bench", bytes(8)))
    d_idx = ctx.index_dev(idx)
    planes_u8 = rng.integers(0, 256, (F, H, W), dtype=np.uint8)
    planes_f = rng.normal(0, 50, (F, H, W)).astype(np.float32)
    d_u8 = ctx.malloc(F * n); ctx.h2d(d_u8, planes_u8)
    d_u8b = ctx.malloc(F * n); ctx.h2d(d_u8b, planes_u8[::-1].copy())
    d_f = ctx.malloc(F * n * 4); ctx.h2d(d_f, planes_f)
    d_g = ctx.malloc(F * n * 4)
    d_bgr = ctx.malloc(F * n * 3); d_bgr2 = ctx.malloc(F * n * 3)
    ctx.h2d(d_bgr, rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8))
    d_s = ctx.malloc(64)
    rows = []

    def timed(name, kernel, fn, alg_bytes, note=""):
        fn(); ctx.sync()
        ctx.event_record(0)
        for _ in range(a.reps):
            fn()
        ctx.event_record(1)
        ms = ctx.event_elapsed_ms(0, 1) / a.reps
        gbs = alg_bytes / ms / 1e6
        rows.append(dict(name=name, kernel=kernel, us_per_launch=ms * 1e3, us_per_plane=ms * 1e3 / F, algorithmic_bytes=alg_bytes,
                         GBps=gbs, frac_of_8TBps=gbs / HBM, note=note))
        print(f"{name:28s} {ms * 1e3:9.1f} us/launch {ms * 1e3 / F:8.1f} us/plane {gbs:8.0f} GB/s = {gbs / HBM * 100:5.1f} % of 8 TB/s  {note}", flush=True)

    vp = api._vp
    call = ctx._call
    timed("permute u8->f32 (gather)", "k_permute<u8>", lambda: call("wm_permute_u8_f32_dev", vp(d_u8), vp(d_idx), vp(d_g), n, F), F * n * 9.0,
          "1 B gathered + 4 B index + 4 B out per px")
    timed("permute f32 (gather)", "k_permute<f32>", lambda: call("wm_permute_f32_dev", vp(d_f), vp(d_idx), vp(d_g), n, F), F * n * 12.0)
    timed("unpermute f32 (scatter)", "k_unpermute", lambda: call("wm_unpermute_f32_dev", vp(d_f), vp(d_idx), vp(d_g), n, F), F * n * 12.0,
          "4 B in + 4 B index + 4 B scattered per px")
    timed("minmax + normalise -> u8", "k_minmax + k_normalize_u8", lambda: [call("wm_normalize_u8_dev", vp(d_f + z * n * 4), n, 1, vp(d_u8b + z * n)) for z in range(F)],
          F * n * 9.0, "4 B (minmax) + 4 B + 1 B per px, one plane per launch pair")
    route = ctx.route_dev(idx)
    timed("permute u8->f32 routed", "k_route_ga + k_route_gb", lambda: call("wm_permute_u8_f32_routed_dev", vp(d_u8), vp(route), vp(d_g), n, F), F * n * 9.0,
          "algorithmic 9 B per px; the routed form moves 13 B, all coalesced")
    timed("unscramble+normalise routed", "k_minmax_planes + k_route_p1 + k_route_p2",
          lambda: call("wm_unpermute_normalize_u8_dev", vp(d_f), vp(route), vp(d_u8b), n, F, 1), F * n * 13.0,
          "algorithmic: 4 B (minmax) + 4 B + 4 B index + 1 B out per px; the routed form moves 17 B, all coalesced")
    timed("bgr -> ycrcb", "k_color<0>", lambda: call("wm_bgr_to_ycrcb_u8_dev", vp(d_bgr), vp(d_bgr2), F * n), F * n * 6.0)
    timed("bgr -> y", "k_color<3>", lambda: call("wm_bgr_to_y_u8_dev", vp(d_bgr), vp(d_u8b), F * n), F * n * 4.0)
    timed("replace y (ycrcb -> bgr)", "k_color<4>", lambda: call("wm_replace_y_u8_dev", vp(d_bgr), vp(d_u8), vp(d_bgr2), F * n), F * n * 7.0)
    timed("squared difference (PSNR)", "k_sqdiff_u8", lambda: call("wm_sqdiff_u8_dev", vp(d_bgr), vp(d_bgr2), F * n * 3, vp(d_s)), F * n * 6.0)
    timed("ssim (u8, u8)", "k_ssim", lambda: [call("wm_ssim_dev", vp(d_u8 + z * n), W, vp(d_u8b + z * n), W, H, W, 0, vp(d_s)) for z in range(F)], F * n * 2.0,
          "VALU-bound: 11x11 Gaussian of 5 fields")
    timed("ssim (u8, f32)", "k_ssim", lambda: [call("wm_ssim_dev", vp(d_u8 + z * n), W, vp(d_f + z * n * 4), W, H, W, 2, vp(d_s)) for z in range(F)], F * n * 5.0)
    ctx.check_status()
    if a.json:
        json.dump(dict(H=H, W=W, frames=F, reps=a.reps, rows=rows), open(a.json, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
