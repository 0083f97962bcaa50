"""max |sigma(sigma-only kernel) - Sc(embed kernel)| / sigma_1 over full frames, per library variant"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
from tools.ab_embed import Ctx
rng = np.random.default_rng(1234)
host = rng.integers(0, 256, (4, 2160, 3840), dtype=np.uint8)
yy, xx = np.mgrid[0:2160, 0:3840]
host[3] = np.clip(128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 2, (2160, 3840)), 0, 255)
S = np.abs(rng.normal(0, 100, (270, 480, 8))).astype(np.float32)
for path in sys.argv[1:]:
    c = Ctx(api.load_library(os.path.abspath(path)))
    _, sc, _ = c.embed_tiles(host, S, 0.0)
    s = c.sigma_tiles(host)
    d = np.abs(s - sc) / sc[..., :1]
    print(os.path.basename(path), "max rel-to-sigma1 deviation per frame:", [f"{d[z].max():.2e}" for z in range(4)], flush=True)
