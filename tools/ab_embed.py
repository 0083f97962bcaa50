"""Interleaved A/B timing of k_embed_tiles / k_extract_tiles builds in ONE process on one device
(cdna_hip_programming.md 5.4 rule 24): every variant is a separate libwmhip build (tools/build_variants.sh),
rounds alternate between them, median and min are reported, and every variant's stego / Sc / extract output
is compared with variant 0's.

    python tools/ab_embed.py tools/bin/libwmhip_base.so tools/bin/libwmhip_skip.so ...
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
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--alpha", type=float, default=0.15)
    a = ap.parse_args()
    H, W, F = a.H, a.W, a.frames
    nt = (H // 8) * (W // 8)
    rng = np.random.default_rng(1234)
    host = rng.integers(0, 256, (F, H, W), dtype=np.uint8)
    wys = rng.integers(0, 256, (H, W)).astype(np.float32)
    V = []
    for path in a.libs:
        lib = api.load_library(os.path.abspath(path))
        c = Ctx(lib)
        d = dict(name=os.path.basename(path).replace("libwmhip_", "").replace(".so", ""), ctx=c)
        d["host"] = c.malloc(host.nbytes); c.h2d(d["host"], host)
        d["stego"] = c.malloc(host.nbytes)
        d["wys"] = c.malloc(wys.nbytes); c.h2d(d["wys"], wys)
        for k, n in (("U", nt * 256), ("V", nt * 256), ("Ux", nt * 256), ("Vx", nt * 256), ("S", nt * 32),
                     ("sc", F * nt * 32), ("out", F * H * W * 4)):
            d[k] = c.malloc(n)
        c.svd_tiles_f32_dev(d["wys"], d["U"], d["S"], d["V"], 1, H, W, W, H * W)
        c.tile_factors_to_pixel_dev(d["U"], d["V"], d["Ux"], d["Vx"], nt)
        d["embed"] = (lambda d=d, c=c: c.embed_tiles_u8_dev(d["host"], d["S"], d["stego"], d["sc"], None, F, H, W, W,
                                                            H * W, 0, a.alpha, 8))
        d["extract"] = (lambda d=d, c=c: c.extract_tiles_px_u8_dev(d["stego"], d["sc"], d["Ux"], d["Vx"], d["out"], F, H,
                                                                   W, W, H * W, 0, a.alpha, 8))
        d["embed"](); d["extract"](); c.sync(); c.check_status()
        d["t"] = dict(embed=[], extract=[])
        V.append(d)
    for _ in range(a.rounds):
        for d in V:
            c = d["ctx"]
            for op in ("embed", "extract"):
                c.event_record(0)
                for _ in range(a.reps):
                    d[op]()
                c.event_record(1)
                d["t"][op].append(c.event_elapsed_ms(0, 1) / a.reps)
    ref = None
    for d in V:
        c = d["ctx"]
        st = np.empty((F, H, W), np.uint8); c.d2h(st, d["stego"])
        sc = np.empty((F, nt, 8), np.float32); c.d2h(sc, d["sc"])
        wm = np.empty((min(2, F), H, W), np.float32); c.d2h(wm, d["out"])
        if ref is None:
            ref = (st, sc, wm)
        dd = np.abs(st.astype(np.int16) - ref[0].astype(np.int16))
        rel = float(np.max(np.abs(sc - ref[1]) / np.maximum(ref[1][..., :1], 1e-30)))
        te, tx = np.array(d["t"]["embed"]), np.array(d["t"]["extract"])
        print(f"{d['name']:14s} embed med {np.median(te)*1e3:8.1f} us min {te.min()*1e3:8.1f} | extract med "
              f"{np.median(tx)*1e3:8.1f} us min {tx.min()*1e3:8.1f} | vs variant 0: stego max {int(dd.max())} LSB on "
              f"{float((dd != 0).mean()):.2e} of px, Sc rel {rel:.1e}, extract max abs {float(np.abs(wm - ref[2]).max()):.2e}",
              flush=True)
        c.check_status()


if __name__ == "__main__":
    main()
