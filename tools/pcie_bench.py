"""End-to-end (PCIe-inclusive) embed throughput: pinned host frames -> H2D -> K1 embed
-> D2H stego + Sc, double-buffered on two HIP streams (torch supplies pinned memory and
streams; the kernels run through the C ABI on those streams).  This is the figure
DESIGN.md quotes next to the device-resident one; it is never bench.py's `value`."""
import argparse, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")

ap = argparse.ArgumentParser()
ap.add_argument("--H", type=int, default=2160); ap.add_argument("--W", type=int, default=3840)
ap.add_argument("--frames", type=int, default=16, help="frames per batch"); ap.add_argument("--batches", type=int, default=16)
a = ap.parse_args()
H, W, F = a.H, a.W, a.frames
nt = (H // 8) * (W // 8)
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(2)]
ctxs = [api.Context(0, stream=s.cuda_stream) for s in streams]
h_in = [torch.randint(0, 256, (F, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
h_out = [torch.empty((F, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
h_sc = [torch.empty((F, nt, 8), dtype=torch.float32).pin_memory() for _ in range(2)]
d_in = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
d_out = [torch.empty_like(d_in[0]) for _ in range(2)]
d_sc = [torch.empty((F, nt, 8), dtype=torch.float32, device=dev) for _ in range(2)]
Sw = torch.rand((nt, 8), device=dev) * 100

def run(n_batches, overlap):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(n_batches):
        k = b & 1 if overlap else 0
        with torch.cuda.stream(streams[k]):
            d_in[k].copy_(h_in[k], non_blocking=True)
            ctxs[k].embed_tiles_u8_dev(d_in[k].data_ptr(), Sw.data_ptr(), d_out[k].data_ptr(), d_sc[k].data_ptr(), None,
                                       F, H, W, W, H * W, 0, 0.15, 8)
            h_out[k].copy_(d_out[k], non_blocking=True)
            h_sc[k].copy_(d_sc[k], non_blocking=True)
        if not overlap:
            streams[k].synchronize()
    torch.cuda.synchronize()
    return n_batches * F / (time.perf_counter() - t0)

run(2, True)
print(f"serial (one stream, sync per batch): {run(a.batches, False):9.0f} frames/s")
print(f"double-buffered on two streams:     {run(a.batches, True):9.0f} frames/s")
mb = (H * W * 2 + nt * 32) / 1e6
print(f"PCIe bytes per frame: {mb:.1f} MB  (H2D {H*W/1e6:.1f}, D2H {H*W/1e6 + nt*32/1e6:.1f})")
