"""One warm full-frame extract at 1080p (host-pointer API) for a kernel trace: where its time goes beyond the stego's SVD."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
H, W = 1080, 1920
host = np.random.default_rng(1234).integers(0, 256, (H, W), dtype=np.uint8)
wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
ctx = api.Context(0)
K, alpha = 648, 0.15
U, S, Vt = ctx.ref_svd(wys, True)
st, sc, _ = ctx.ref_embed(host, S, alpha, K)
for i in range(3):
    t0 = time.perf_counter(); w = ctx.ref_extract(st, sc, U, Vt, alpha, K); t1 = time.perf_counter()
    print(f"extract call {i}: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); s = ctx.ref_sigma(st); print(f"sigma: {(time.perf_counter() - t0) * 1e3:.1f} ms")
