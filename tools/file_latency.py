"""File-level latency of the drop-in (PNG in, PNG + .npz out), the call shape of the reference's apps."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_svd_core_secure as core
from PIL import Image

rng = np.random.default_rng(1)
d = tempfile.mkdtemp()
HH, WW = (2160, 3840) if "--4k" in sys.argv else (1080, 1920)
yy, xx = np.mgrid[0:HH, 0:WW]
cover = np.clip(128 + 70 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + rng.normal(0, 6, (HH, WW, 3)), 0, 255).astype(np.uint8)
Image.fromarray(cover).save(os.path.join(d, "cover.png"), compress_level=1)
Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(os.path.join(d, "wm.png"))
for tile, color, cm in ((8, False, True), (8, False, False), (8, True, True), (8, True, False), (None, False, True), (None, True, True)):
    if True:
        for rep in range(2):
            t0 = time.perf_counter()
            out, meta, ps, ss = core.embed(os.path.join(d, "cover.png"), os.path.join(d, "wm.png"), os.path.join(d, "s.png"),
                                           os.path.join(d, "m.npz"), alpha=0.12, color=color, password="pw", tile=tile,
                                           compress_meta=cm)
            t1 = time.perf_counter()
            core.extract(out, meta, os.path.join(d, "w.png"), password="pw")
            t2 = time.perf_counter()
            ok, score = core.detect(out, meta)
            t3 = time.perf_counter()
        print(f"{HH}p tile={tile} color={color} compress_meta={cm}: embed {1e3*(t1-t0):7.1f} ms  extract {1e3*(t2-t1):7.1f} ms  detect {1e3*(t3-t2):7.1f} ms  "
              f"psnr {ps:.2f} ssim {ss:.4f} score {score:.3f}", flush=True)
