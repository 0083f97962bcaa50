"""Array-level latency of the drop-in (decoded images in, arrays out): embed / extract / detect,
gray and colour, tile=8 and tile=None, on one 1080p and one 4K cover."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_svd_core_secure as core

rng = np.random.default_rng(1)
wm = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
for (H, W) in ((1080, 1920), (2160, 3840)):
    cover = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for tile in (8, None):
        for color in (False, True):
            if tile is None and (H > 1080):
                continue
            core.embed_arrays(cover, wm, "pw", bytes(8), alpha=0.12, color=color, tile=tile)      # warm-up
            best = [1e9, 1e9, 1e9]
            for _ in range(3):                                                                      # best of three
                t0 = time.perf_counter(); r = core.embed_arrays(cover, wm, "pw", bytes(8), alpha=0.12, color=color, tile=tile); t1 = time.perf_counter()
                w = core.extract_arrays(r["stego"], r["meta"], "pw"); t2 = time.perf_counter()
                ok, sc = core.detect_arrays(r["stego"], r["meta"]); t3 = time.perf_counter()
                best = [min(best[0], t1 - t0), min(best[1], t2 - t1), min(best[2], t3 - t2)]
            print(f"{W}x{H} tile={tile} color={color}: embed {1e3*best[0]:7.1f} ms  extract {1e3*best[1]:7.1f} ms  detect {1e3*best[2]:7.1f} ms  score {sc:.3f}", flush=True)
