import cProfile, pstats, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_svd_core_secure as core
rng = np.random.default_rng(1)
wm = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
cover = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
r = core.embed_arrays(cover, wm, "pw", bytes(8), alpha=0.12)
pr = cProfile.Profile(); pr.enable()
r = core.embed_arrays(cover, wm, "pw", bytes(8), alpha=0.12)
w = core.extract_arrays(r["stego"], r["meta"], "pw")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
