"""Scale drift of the full-frame Jacobi's rotated rows and its calibration: sigma error against float64
LAPACK with WM_RF_DRIFT_CAL=0 / 1, and the spread of the measured drift factor (WM_RF_DEBUG_DRIFT=1, stderr).
    python tools/ff_drift_debug.py [sizes: 1080 2160 4320]"""
import importlib, os, sys, subprocess, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = {"1080": (1080, 1920), "2160": (2160, 3840), "4320": (4320, 7680), "512": (512, 512)}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
    ctx = api.Context(0)
    for name in sys.argv[2:]:
        H, W = SIZES[name]
        x = np.random.default_rng(1234).integers(0, 256, (H, W), dtype=np.uint8)
        cache = f"/tmp/ff_drift_ref_{name}.npy"
        if os.path.exists(cache):
            ref = np.load(cache)
        else:
            ref = np.linalg.svd(x.astype(np.float64), compute_uv=False); np.save(cache, ref)
        s = ctx.ref_sigma(x).astype(np.float64)
        t0 = time.perf_counter(); s = ctx.ref_sigma(x).astype(np.float64); dt = time.perf_counter() - t0
        e = np.abs(s - ref); i = int(e.argmax())
        print(json.dumps(dict(size=name, cal=os.environ.get("WM_RF_DRIFT_CAL", "1"), sweeps=ctx.ref_last_sweeps(), ms=round(dt * 1e3, 1),
                              max_err_s1=float(e[i] / ref[0]), at=i, ratio_at=float(ref[i] / ref[0]),
                              max_rel_big=float(np.max((e / ref)[ref > 1.1e-2 * ref[0]])),
                              max_rel_small=float(np.max((e / ref)[ref <= 0.9e-2 * ref[0]])))), flush=True)
else:
    sizes = sys.argv[1:] or ["1080", "2160"]
    for cal in ("0", "1"):
        r = subprocess.run([sys.executable, __file__, "child"] + sizes, env=dict(os.environ, WM_RF_DRIFT_CAL=cal, WM_RF_DEBUG_DRIFT="1"),
                           capture_output=True, text=True)
        print(r.stdout.strip()); print(r.stderr.strip()[-1500:], flush=True)
