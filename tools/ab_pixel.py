"""Interleaved A/B timing of the colour kernels (k_color: BGR -> YCrCb, BGR -> Y, replace-Y) of several libwmhip builds in ONE
process; outputs of every variant are compared byte for byte with variant 0's.
    python tools/ab_pixel.py tools/bin/libwmhip_base.so tools/bin/libwmhip_cl.so"""
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
    ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--odd", type=int, default=0, help="pixels taken off the end (tail paths)")
    a = ap.parse_args()
    n = a.H * a.W * a.frames - a.odd
    rng = np.random.default_rng(3)
    bgr = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    ypl = rng.integers(0, 256, n, dtype=np.uint8)
    fpl = rng.normal(0, 50, n).astype(np.float32)
    hg = importlib.import_module(api.__name__.rsplit(".", 1)[0] + ".hostglue")
    npl = a.H * a.W
    idx = hg.permutation_index(a.H, a.W, hg.derive_key("bench", bytes(8))) if a.odd == 0 else None
    vp = api._vp
    V = []
    for path in a.libs:
        lib = api.load_library(os.path.abspath(path))
        c = Ctx(lib)
        d = dict(name=os.path.basename(path).replace("libwmhip_", "").replace(".so", ""), ctx=c)
        d["bgr"] = c.malloc(n * 3); c.h2d(d["bgr"], bgr)
        d["y"] = c.malloc(n); c.h2d(d["y"], ypl)
        d["o3"] = c.malloc(n * 3); d["o1"] = c.malloc(n)
        d["ops"] = dict(
            bgr2ycc=(lambda d=d, c=c: c._call("wm_bgr_to_ycrcb_u8_dev", vp(d["bgr"]), vp(d["o3"]), n), 6.0 * n, "o3"),
            bgr2y=(lambda d=d, c=c: c._call("wm_bgr_to_y_u8_dev", vp(d["bgr"]), vp(d["o1"]), n), 4.0 * n, "o1"),
            replace_y=(lambda d=d, c=c: c._call("wm_replace_y_u8_dev", vp(d["bgr"]), vp(d["y"]), vp(d["o3"]), n), 7.0 * n, "o3"))
        d["f"] = c.malloc(n * 4); c.h2d(d["f"], fpl)
        d["ops"]["minmax_norm"] = (lambda d=d, c=c: [c._call("wm_normalize_u8_dev", vp(d["f"] + z * npl * 4), npl, 1, vp(d["o1"] + z * npl))
                                                     for z in range(max(1, n // npl))], 9.0 * (n // npl) * npl, "o1")
        if idx is not None:
            d["route"] = c.route_dev(idx)
            d["ops"]["unscr_norm_routed"] = (lambda d=d, c=c: c._call("wm_unpermute_normalize_u8_dev", vp(d["f"]), vp(d["route"]), vp(d["o1"]), npl, a.frames, 1),
                                             13.0 * n, "o1")
        d["t"] = {k: [] for k in d["ops"]}
        V.append(d)
    ref = {}
    for d in V:                                   # outputs first (each op overwrites o3)
        c = d["ctx"]
        for k, (fn, _, buf) in d["ops"].items():
            fn(); c.sync()
            out = np.empty(n * (3 if buf == "o3" else 1), np.uint8); c.d2h(out, d[buf])
            if k not in ref:
                ref[k] = out
            d.setdefault("same", {})[k] = bool(np.array_equal(out, ref[k]))
    for _ in range(a.rounds):
        for d in V:
            c = d["ctx"]
            for k, (fn, _, _) in d["ops"].items():
                c.event_record(0)
                for _ in range(a.reps):
                    fn()
                c.event_record(1)
                d["t"][k].append(c.event_elapsed_ms(0, 1) / a.reps)
    for d in V:
        parts = []
        for k, (_, nbytes, _) in d["ops"].items():
            t = np.median(d["t"][k])
            parts.append(f"{k} {t * 1e3:7.1f} us = {nbytes / t / 1e6:6.0f} GB/s ({nbytes / t / 1e6 / 80:4.1f} %) same={d['same'][k]}")
        print(f"{d['name']:12s} " + " | ".join(parts), flush=True)
        d["ctx"].check_status()


if __name__ == "__main__":
    main()
