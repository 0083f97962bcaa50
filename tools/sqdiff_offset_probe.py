import importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
ctx = api.Context(0); vp = api._vp
n = 8 * 2160 * 3840 * 3
d_a = ctx.malloc(n + (4 << 20)); d_b = ctx.malloc(n + (4 << 20)); d_s = ctx.malloc(64)
ctx.memset(d_a, 3, n + (4 << 20)); ctx.memset(d_b, 7, n + (4 << 20))
print("base addresses", hex(d_a), hex(d_b), "delta", hex(d_b - d_a))
for off in (0, 256, 1024, 4096, 4096 + 256, 65536, 1 << 20, (1 << 20) + 4096 + 256):
    f = lambda: ctx._call("wm_sqdiff_u8_dev", vp(d_a), vp(d_b + off), n, vp(d_s))
    f(); ctx.sync(); ctx.event_record(0)
    for _ in range(10): f()
    ctx.event_record(1)
    ms = ctx.event_elapsed_ms(0, 1) / 10
    print(f"b offset {off:9d}: {ms * 1e3:7.1f} us  {2 * n / ms / 1e6:7.0f} GB/s")
