"""Where does the watermark-side SVD's sigma error come from - the DCT GEMMs or the decomposition?  (round 3)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wm_oracle as o
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
ctx = api.Context(0)
for H, W in ((1080, 1920), (2160, 3840)):
    wm = np.random.default_rng(4321).integers(0, 256, (H, W), dtype=np.uint8).astype(np.float32)
    C64 = __import__("scipy.fft").fft.dctn(wm.astype(np.float64), type=2, norm="ortho")
    ref = np.linalg.svd(C64, compute_uv=False)
    U, S, Vt = ctx.ref_svd(wm, apply_dct=True)
    e = np.abs(S - ref)
    print(H, W, "dct on GPU : s1 err %.2e  max err/s1 %.2e  max rel %.2e" % (e[0] / ref[0], e.max() / ref[0], (e / ref).max()))
    R = (U * S) @ Vt
    print("   reconstruction vs float64 DCT: C00 rel %.2e, max abs %.3e (C00 %.1f)" % (abs(R[0, 0] - C64[0, 0]) / C64[0, 0], np.abs(R - C64).max(), C64[0, 0]))
    C32 = C64.astype(np.float32)
    U, S, Vt = ctx.ref_svd(C32, apply_dct=False)
    e = np.abs(S - ref)
    print("   dct on host: s1 err %.2e  max err/s1 %.2e  max rel %.2e" % (e[0] / ref[0], e.max() / ref[0], (e / ref).max()))
    U, S, Vt = ctx.ref_svd(wm, apply_dct=False)
    e = np.abs(S - ref)
    print("   no dct     : s1 err %.2e  max err/s1 %.2e  max rel %.2e" % (e[0] / ref[0], e.max() / ref[0], (e / ref).max()))
    s2 = ctx.ref_sigma(wm.astype(np.uint8)); e = np.abs(s2 - ref)
    print("   sigma-only : s1 err %.2e  max err/s1 %.2e  max rel %.2e" % (e[0] / ref[0], e.max() / ref[0], (e / ref).max()))
