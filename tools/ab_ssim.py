"""Interleaved A/B timing of k_ssim builds in ONE process on one device (cdna_hip_programming.md 5.4 rule 24): every
variant is a separate libwmhip build (tools/build_variants.sh), rounds alternate between them, median and min are
reported, and every variant's mean SSIM (uint8/uint8, uint8/float32, odd sizes, tiny planes) is compared with variant 0's.

    python tools/ab_ssim.py tools/bin/libwmhip_base.so tools/bin/libwmhip_pipe3.so ...
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module(
    "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")


class Ctx(api.Context):
    def __init__(self, lib):
        self.lib = lib
        h = api._vp()
        rc = lib.wm_create(0, None, api.C.byref(h))
        assert rc == 0, lib.wm_last_error()
        self._h = h
        self.device = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--H", type=int, default=2160)
    ap.add_argument("--W", type=int, default=3840)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    H, W, F = a.H, a.W, a.frames
    n = H * W
    rng = np.random.default_rng(99)
    xs = rng.integers(0, 256, (F, H, W), dtype=np.uint8)
    ys = np.clip(xs.astype(np.int16) + rng.integers(-12, 13, xs.shape), 0, 255).astype(np.uint8)
    yf = (ys.astype(np.float32) + rng.normal(0, 0.3, ys.shape).astype(np.float32))
    smalls = [(7, 5), (11, 64), (64, 11), (33, 129), (130, 77), (3, 3), (1, 9), (256, 384)]
    vp = api._vp
    V = []
    for path in a.libs:
        lib = api.load_library(os.path.abspath(path))
        c = Ctx(lib)
        d = dict(name=os.path.basename(path).replace("libwmhip_", "").replace(".so", ""), ctx=c)
        d["x"] = c.malloc(xs.nbytes); c.h2d(d["x"], xs)
        d["y"] = c.malloc(ys.nbytes); c.h2d(d["y"], ys)
        d["yf"] = c.malloc(yf.nbytes); c.h2d(d["yf"], yf)
        d["s"] = c.malloc(8 * F)
        d["u8"] = (lambda d=d, c=c: [c._call("wm_ssim_dev", vp(d["x"] + z * n), W, vp(d["y"] + z * n), W, H, W, 0, vp(d["s"] + 8 * z))
                                     for z in range(F)])
        d["f32"] = (lambda d=d, c=c: [c._call("wm_ssim_dev", vp(d["x"] + z * n), W, vp(d["yf"] + 4 * z * n), W, H, W, 2, vp(d["s"] + 8 * z))
                                      for z in range(F)])
        d["u8"](); d["f32"](); c.sync(); c.check_status()
        d["t"] = dict(u8=[], f32=[])
        V.append(d)
    for _ in range(a.rounds):
        for d in V:
            c = d["ctx"]
            for op in ("u8", "f32"):
                c.event_record(0)
                for _ in range(a.reps):
                    d[op]()
                c.event_record(1)
                d["t"][op].append(c.event_elapsed_ms(0, 1) / a.reps / F)
    ref = None
    for d in V:
        c = d["ctx"]
        vals = []
        d["u8"](); v = np.zeros(F); c.d2h(v, d["s"]); vals += list(v)
        d["f32"](); v = np.zeros(F); c.d2h(v, d["s"]); vals += list(v)
        r2 = np.random.default_rng(5)
        for (h, w) in smalls:
            p = r2.integers(0, 256, (h, w), dtype=np.uint8)
            q = np.clip(p.astype(np.int16) + r2.integers(-20, 21, p.shape), 0, 255).astype(np.uint8)
            vals.append(c.ssim(p, q)); vals.append(c.ssim(p, q.astype(np.float32) + 0.25)); vals.append(c.ssim(q.astype(np.float32), p))
        vals = np.array(vals)
        if ref is None:
            ref = vals
        tu, tf = np.array(d["t"]["u8"]), np.array(d["t"]["f32"])
        print(f"{d['name']:16s} u8/u8 med {np.median(tu)*1e3:7.1f} us/plane min {tu.min()*1e3:7.1f} | u8/f32 med {np.median(tf)*1e3:7.1f} "
              f"min {tf.min()*1e3:7.1f} | max |ssim - variant 0| {float(np.abs(vals - ref).max()):.2e} (ssim[0] {vals[0]:.6f})", flush=True)
        c.check_status()


if __name__ == "__main__":
    main()
