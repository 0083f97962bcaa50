import importlib, os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
    ctx = api.Context(0)
    rng = np.random.default_rng(12)
    out = {}
    for name, H, W in (("noise1080", 1080, 1920), ("smooth256", 256, 384), ("smooth720", 720, 1280)):
        if name.startswith("noise"):
            x = rng.integers(0, 256, (H, W), dtype=np.uint8)
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            x = np.clip(128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 40 * np.sin((xx + 2 * yy) / 91.0) + rng.normal(0, 2, (H, W)), 0, 255).astype(np.uint8)
        ref = np.linalg.svd(x.astype(np.float64), compute_uv=False)
        s = ctx.ref_sigma(x).astype(np.float64); sw = ctx.ref_last_sweeps()
        Sw = np.sort(rng.uniform(10, 3000, min(H, W)).astype(np.float32))[::-1].copy()
        st, sc, _ = ctx.ref_embed(x, Sw, 0.15, int(0.6 * min(H, W)))
        sc = sc.astype(np.float64)
        out[name] = dict(sweeps=sw, sig_rel=float(np.max(np.abs(s - ref) / ref)), sig_s1=float(np.max(np.abs(s - ref)) / ref[0]),
                         sc_rel=float(np.max(np.abs(sc - ref) / ref)), sc_s1=float(np.max(np.abs(sc - ref)) / ref[0]),
                         med=float(np.median(np.abs(s - ref) / ref)))
    print(json.dumps(out))
else:
    for pol in ("0", "1", "2"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, WM_RF_SIGMA_POLICY=pol), capture_output=True, text=True)
        print("policy", pol, r.stdout.strip() or r.stderr[-400:], flush=True)
