"""The PCIe-inclusive pipeline of bench.py (end_to_end_section) on its own, for a copy / kernel timeline:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/xx -- python3 tools/e2e_probe.py
and variants that take stages out (--no-kernels, --no-h2d, --no-d2h) to see which stage holds the others up."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
PKG = bench.PKG


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=15); ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--nbuf", type=int, default=3)
    ap.add_argument("--no-main-ctx", action="store_true", help="random factors instead of a context with its own stream doing the watermark SVD")
    a = ap.parse_args()
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    api = importlib.import_module(PKG + ".hostapi"); hg = importlib.import_module(PKG + ".hostglue")
    H, W = 2160, 3840; nt = (H // 8) * (W // 8)
    if a.no_main_ctx:
        Sw = torch.rand((nt, 8), dtype=torch.float32, device=dev) * 100; Ux = torch.rand((nt, 8, 8), dtype=torch.float32, device=dev); Vxt = Ux.clone()
    else:
        ctx = api.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        Sw = torch.zeros((nt, 8), dtype=torch.float32, device=dev); Uw = torch.zeros((nt, 8, 8), dtype=torch.float32, device=dev); Vwt = torch.zeros_like(Uw)
        wys = torch.from_numpy(np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)).to(dev)
        ctx.svd_tiles_f32_dev(wys.data_ptr(), Uw.data_ptr(), Sw.data_ptr(), Vwt.data_ptr(), 1, H, W, W, H * W)
        Ux = torch.empty_like(Uw); Vxt = torch.empty_like(Vwt)
        ctx.tile_factors_to_pixel_dev(Uw.data_ptr(), Vwt.data_ptr(), Ux.data_ptr(), Vxt.data_ptr(), nt)
        torch.cuda.synchronize(dev)
    idx = hg.permutation_index(H, W, hg.derive_key("bench", bytes(8)))
    pre = os.environ.get("WM_PROBE_PRE", "")
    if pre == "roof":
        bench.pcie_roof(torch, dev, 8 * H * W, 8 * (2 * H * W + nt * 32))
    if pre == "pinned":
        t = [torch.empty(8 * (2 * H * W + nt * 32), dtype=torch.uint8).pin_memory() for _ in range(3)] + [torch.empty((8, H, W), dtype=torch.uint8).pin_memory() for _ in range(3)]
        del t
    if pre == "streams":
        ss = [torch.cuda.Stream(dev) for _ in range(3)]
        x = torch.zeros(1 << 20, device=dev)
        for q in ss:
            with torch.cuda.stream(q):
                x.add_(1)
        torch.cuda.synchronize()
    r = bench.end_to_end_section(torch, api, dev, H, W, 0.15, Sw, Ux, Vxt, idx, F=a.frames, batches=a.batches, nbuf=a.nbuf)
    if os.environ.get("WM_PROBE_MID") == "empty":
        torch.cuda.empty_cache()
        if hasattr(torch._C, "_host_emptyCache"):
            torch._C._host_emptyCache(); print("host cache emptied", file=sys.stderr)
    r2 = bench.end_to_end_section(torch, api, dev, H, W, 0.15, Sw, Ux, Vxt, idx, F=a.frames, batches=a.batches, nbuf=a.nbuf)
    r["second_call"] = [r2["value"], r2["stream_sets_tried_short_run_frames_per_s"]]
    print(json.dumps(r))


if __name__ == "__main__":
    main()
