import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
ctx = api.Context(0)
for H, W in ((128, 128), (256, 384), (512, 512), (1080, 1920)):
    x = np.random.default_rng(7).integers(0, 256, (H, W), dtype=np.uint8)
    s = ctx.ref_sigma(x).astype(np.float64)
    ref = np.linalg.svd(x.astype(np.float64), compute_uv=False)
    e = np.abs(s - ref)
    i = int(np.argmax(e)); j = int(np.argmax(e / ref))
    print(H, W, "sweeps", ctx.ref_last_sweeps(), "s1 gpu/ref", s[0], ref[0], "rel", e[0] / ref[0],
          "| worst abs idx", i, e[i], ref[i], "| worst rel idx", j, e[j] / ref[j], ref[j],
          "| median rel", float(np.median(e / ref)))
