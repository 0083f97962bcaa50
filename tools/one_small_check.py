"""Flagged full-rank tiles of noise frames (s_8 < 1e-5 s_1): the one-small completion (default build) and the literal chain
(-DWM_EXP_NO_ONE_SMALL build) against float64 LAPACK, tile by tile.
    python tools/one_small_check.py tools/bin/libwmhip_nosmall.so tools/bin/libwmhip_small.so"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
H, W, F = 2160, 3840, int(os.environ.get("FRAMES", "4"))
nby, nbx = H // 8, W // 8
rng = np.random.default_rng(1234)
host = rng.integers(0, 256, (F, H, W), dtype=np.uint8)
sw = np.sort(rng.uniform(1, 2000, (nby, nbx, 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
tiles = host.reshape(F, nby, 8, nbx, 8).transpose(0, 1, 3, 2, 4).astype(np.float64)
U, S, Vt = np.linalg.svd(tiles)
flag = S[..., 7] < 1e-5 * S[..., 0]
print("flagged tiles:", int(flag.sum()), "of", flag.size, " with s_7 also < 1e-4 s_1:", int((flag & (S[..., 6] < 1e-4 * S[..., 0])).sum()))
ref = (U[flag] * (S[flag] + 0.15 * np.broadcast_to(sw, S.shape)[flag])[:, None, :]) @ Vt[flag]
for path in sys.argv[1:]:
    lib = api.load_library(os.path.abspath(path))
    c = api.Context.__new__(api.Context); c.lib = lib
    h = api._vp(); assert lib.wm_create(0, None, api.C.byref(h)) == 0; c._h = h; c.device = 0
    st, sc, yw = c.embed_tiles(host, sw, 0.15, want_yw=True)
    T = yw.reshape(F, nby, 8, nbx, 8).transpose(0, 1, 3, 2, 4)[flag].astype(np.float64)
    d = np.abs(T - ref).max(axis=(1, 2))
    q = np.abs(np.clip(T, 0, 255).astype(np.uint8).astype(int) - np.clip(ref, 0, 255).astype(np.uint8).astype(int)).max(axis=(1, 2))
    worst = int(d.argmax())
    print(f"{os.path.basename(path):28s} Yw vs LAPACK on the flagged tiles: max {d.max():.3f} grey levels (median {np.median(d):.4f}), stego max {int(q.max())} LSB, "
          f"tiles off by > 1 LSB: {int((q > 1).sum())};  worst tile s_7/s_1 {S[flag][worst, 6] / S[flag][worst, 0]:.2e}, s_8/s_1 {S[flag][worst, 7] / S[flag][worst, 0]:.2e}")
    allT = yw.reshape(F, nby, 8, nbx, 8).transpose(0, 1, 3, 2, 4)[~flag].astype(np.float64)
    refall = (U[~flag] * (S[~flag] + 0.15 * np.broadcast_to(sw, S.shape)[~flag])[:, None, :]) @ Vt[~flag]
    qa = np.abs(np.clip(allT, 0, 255).astype(np.uint8).astype(int) - np.clip(refall, 0, 255).astype(np.uint8).astype(int))
    print(f"{'':28s} all other tiles: stego max {int(qa.max())} LSB on {float((qa != 0).mean()):.2e} of the pixels")
