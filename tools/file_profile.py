"""cProfile of one file-level embed + extract (1080p gray, tile=8): where the host time of the drop-in goes."""
import cProfile, os, pstats, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_svd_core_secure as core
from PIL import Image
COLOR = "--color" in sys.argv
rng = np.random.default_rng(1)
d = tempfile.mkdtemp()
HH, WW = (2160, 3840) if "--4k" in sys.argv else (1080, 1920)
yy, xx = np.mgrid[0:HH, 0:WW]
cover = np.clip(128 + 70 * np.sin(xx / 37.0)[..., None] * np.cos(yy / 23.0)[..., None] + rng.normal(0, 6, (HH, WW, 3)), 0, 255).astype(np.uint8)
Image.fromarray(cover).save(os.path.join(d, "cover.png"), compress_level=1)
Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(os.path.join(d, "wm.png"))
args = (os.path.join(d, "cover.png"), os.path.join(d, "wm.png"), os.path.join(d, "s.png"), os.path.join(d, "m.npz"))
core.embed(*args, alpha=0.12, password="pw", color=COLOR)
pr = cProfile.Profile()
if "--embed" in sys.argv:
    pr.enable()
out, meta, ps, ss = core.embed(*args, alpha=0.12, password="pw", color=COLOR)
pr.enable()
core.extract(out, meta, os.path.join(d, "w.png"), password="pw")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
