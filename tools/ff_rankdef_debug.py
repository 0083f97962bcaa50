"""Full-frame mode on rank-deficient planes: sweeps, sigma vs float64 LAPACK, embed sanity."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
ctx = api.Context(0)
rng = np.random.default_rng(5)
cases = {}
x = rng.integers(0, 256, (128, 72), dtype=np.uint8); x[:64] = x[:1, :1]; cases["half constant 128x72"] = x
x = rng.integers(0, 256, (96, 160), dtype=np.uint8); x[20:60] = x[20]; cases["40 equal rows 96x160"] = x
cases["flat 64x64"] = np.full((64, 64), 90, np.uint8)
x = rng.integers(0, 256, (1080, 1920), dtype=np.uint8); x[:140] = 16; x[-140:] = 16; cases["letterbox 1080p"] = x
yy, xx = np.mgrid[0:256, 0:384]; cases["smooth synthetic 256x384"] = (128 + 100 * np.sin(xx / 40.0) * np.cos(yy / 30.0)).astype(np.uint8)
for name, x in cases.items():
    H, W = x.shape
    try:
        s = ctx.ref_sigma(x).astype(np.float64)
    except Exception as e:
        print(name, "sigma FAILED:", e); continue
    sw = ctx.ref_last_sweeps()
    ref = np.linalg.svd(x.astype(np.float64), compute_uv=False)
    e = np.abs(s - ref)
    print(f"{name}: sweeps {sw}, rank(1e-6) {int((ref > 1e-6 * ref[0]).sum())}/{len(ref)}, max err/s1 {e.max() / ref[0]:.2e}, "
          f"worst idx {int(e.argmax())} gpu {s[int(e.argmax())]:.4g} ref {ref[int(e.argmax())]:.4g}")
    Sw = np.sort(rng.uniform(10, 3000, min(H, W)).astype(np.float32))[::-1].copy()
    st, sc, yw = ctx.ref_embed(x, Sw, 0.15, int(0.6 * min(H, W)), want_yw=True)
    print(f"   embed: sweeps {ctx.ref_last_sweeps()}, finite {np.isfinite(yw).all()}, |yw - x| max {np.abs(yw - x).max():.2f}, sc err/s1 {np.abs(sc - ref).max() / ref[0]:.2e}")
