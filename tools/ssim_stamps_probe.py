"""k_ssim with in-kernel s_memtime stamps (a -DWM_SSIM_STAMPS build: `tools/build_variants.sh stamps:"-DWM_SSIM_STAMPS"`, add
`-DWM_SSIM_ROTPRIO=0` for the arbiter's own oldest-first order): per sampled workgroup its dispatch rank, lifetime and the
cycles per row up to the ring write (E) and from there to the next row (D).  Output: the kernel's printf lines."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
lib = api.load_library(os.path.abspath("tools/bin/libwmhip_stamps.so"))
class Ctx(api.Context):
    def __init__(self, lib):
        self.lib = lib; h = api._vp(); assert lib.wm_create(0, None, api.C.byref(h)) == 0; self._h = h; self.device = 0
c = Ctx(lib)
rng = np.random.default_rng(1)
for (H, W) in ((2160, 3840),):
    a = rng.integers(0, 256, (H, W), dtype=np.uint8); b = rng.integers(0, 256, (H, W), dtype=np.uint8)
    print(f"== {H} x {W}: {((W+63)//64)*((H+33)//34)/1024:.2f} waves per SIMD", flush=True)
    for _ in range(2):
        v = c.ssim(a, b); c.sync(); print('-- launch done', flush=True)
    c.sync()
