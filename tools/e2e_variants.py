"""Which stage of the three-stream PCIe pipeline holds the others up?  Variants of bench.end_to_end_section's loop
(4K frames, F per batch): copies only, kernels replaced by nothing, one D2H copy instead of three, issue order.
    python tools/e2e_variants.py"""
import importlib, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
PKG = bench.PKG
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
api = importlib.import_module(PKG + ".hostapi"); hg = importlib.import_module(PKG + ".hostglue")
H, W, F, nbuf = 2160, 3840, 8, 3
nt = (H // 8) * (W // 8); n = H * W
s_up, s_k, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ctx = api.Context(0, stream=s_k.cuda_stream)
Sw = torch.rand((nt, 8), dtype=torch.float32, device=dev) * 100; Ux = torch.rand((nt, 8, 8), dtype=torch.float32, device=dev); Vxt = Ux.clone()
idx = hg.permutation_index(H, W, hg.derive_key("bench", bytes(8)))
route = ctx.route_dev(idx)
h_in = [torch.randint(0, 256, (F, H, W), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
out_bytes = F * (2 * n + nt * 32)
h_out = [torch.empty(out_bytes, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
d_in = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
d_out = [torch.empty(out_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)]     # stego | wm u8 | Sc, one D2H
d_wm = torch.empty((F, H, W), dtype=torch.float32, device=dev)


SEP = {"on": False}
d_st_s = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
d_u8_s = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
d_sc_s = [torch.empty((F, nt, 8), dtype=torch.float32, device=dev) for _ in range(nbuf)]
h_st_s = [torch.empty((F, H, W), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
h_u8_s = [torch.empty((F, H, W), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
h_sc_s = [torch.empty((F, nt, 8), dtype=torch.float32).pin_memory() for _ in range(nbuf)]
d_sc_b = [torch.empty(F * nt * 32, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
h_sc_b = [torch.empty(F * nt * 32, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]


def views(k):
    if SEP["on"]:
        return d_st_s[k], d_u8_s[k], (d_sc_b[k] if SEP.get("mode") == "u8sc" else d_sc_s[k])
    st = d_out[k][:F * n]; u8 = d_out[k][F * n:2 * F * n]; sc = d_out[k][2 * F * n:]
    return st, u8, sc


ctx_up = api.Context(0, stream=s_up.cuda_stream); ctx_dn = api.Context(0, stream=s_dn.cuda_stream)


def run(nb, kernels=True, h2d=True, d2h=True, split_d2h=False, events=True, kcopy_d2h=0, kcopy_h2d=0):
    ev_up = [torch.cuda.Event() for _ in range(nbuf)]; ev_k = [torch.cuda.Event() for _ in range(nbuf)]
    ev_free = [torch.cuda.Event() for _ in range(nbuf)]; ev_dn = [torch.cuda.Event() for _ in range(nbuf)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for b in range(nb):
        k = b % nbuf
        st, u8, sc = views(k)
        with torch.cuda.stream(s_up):
            if events and b >= nbuf:
                s_up.wait_event(ev_free[k])
            if h2d and kcopy_h2d:
                ctx_up.copy_mapped(d_in[k].data_ptr(), h_in[k].data_ptr(), F * n, kcopy_h2d)
            elif h2d:
                d_in[k].copy_(h_in[k], non_blocking=True)
            ev_up[k].record(s_up)
        with torch.cuda.stream(s_k):
            if events:
                s_k.wait_event(ev_up[k])
                if b >= nbuf:
                    s_k.wait_event(ev_dn[k])
            if kernels:
                ctx.embed_tiles_u8_dev(d_in[k].data_ptr(), Sw.data_ptr(), st.data_ptr(), sc.data_ptr(), None, F, H, W, W, H * W, 0, 0.15, 8)
            ev_free[k].record(s_k)
            if kernels:
                ctx.extract_tiles_px_u8_dev(st.data_ptr(), sc.data_ptr(), Ux.data_ptr(), Vxt.data_ptr(), d_wm.data_ptr(), F, H, W, W, H * W, 0, 0.15, 8)
                ctx._call("wm_unpermute_normalize_u8_dev", api._vp(d_wm.data_ptr()), api._vp(route), api._vp(u8.data_ptr()), n, F, 1)
            ev_k[k].record(s_k)
        with torch.cuda.stream(s_dn):
            if events:
                s_dn.wait_event(ev_k[k])
            if d2h and kcopy_d2h:
                ctx_dn.copy_mapped(h_out[k].data_ptr(), d_out[k].data_ptr(), out_bytes, kcopy_d2h)
            elif d2h:
                if SEP["on"]:
                    m = SEP.get("mode")
                    h_st_s[k].copy_(d_st_s[k], non_blocking=True)
                    if m == "u8sc":
                        h_sc_b[k].copy_(d_sc_b[k], non_blocking=True)
                    elif m == "sc_view":
                        h_sc_s[k].view(torch.uint8).copy_(d_sc_s[k].view(torch.uint8), non_blocking=True)
                    elif m == "sc_last":
                        pass
                    elif m != "nosc":
                        h_sc_s[k].copy_(d_sc_s[k], non_blocking=True)
                    h_u8_s[k].copy_(d_u8_s[k], non_blocking=True)
                    if m == "sc_last":
                        h_sc_s[k].copy_(d_sc_s[k], non_blocking=True)
                elif split_d2h:
                    a, c = F * n, 2 * F * n
                    h_out[k][:a].copy_(d_out[k][:a], non_blocking=True)
                    h_out[k][a:c].copy_(d_out[k][a:c], non_blocking=True)
                    h_out[k][c:].copy_(d_out[k][c:], non_blocking=True)
                else:
                    h_out[k].copy_(d_out[k], non_blocking=True)
            ev_dn[k].record(s_dn)
    torch.cuda.synchronize(dev)
    return nb * F / (time.perf_counter() - t0)


if os.environ.get("WM_VAR_ONLY"):
    kw = eval(os.environ["WM_VAR_ONLY"])
    print("only", kw, [round(run(15, **kw)) for _ in range(8)], flush=True)
    sys.exit(0)
SEP["on"] = True
for m in (None, "nosc", "u8sc", "sc_view", "sc_last"):
    SEP["mode"] = m
    run(3); print(f"{'separate tensors, Sc copy variant ' + str(m):45s} {run(15):9.0f} frames/s", flush=True)
SEP["on"] = False
for name, kw in (("D2H by copy kernel, 64 workgroups", dict(kcopy_d2h=64)),
                 ("D2H by copy kernel, 16 workgroups", dict(kcopy_d2h=16)),
                 ("D2H by copy kernel, 256 workgroups", dict(kcopy_d2h=256)),
                 ("both directions by copy kernel (64 / 64)", dict(kcopy_d2h=64, kcopy_h2d=64)),
                 ("both by copy kernel, copies only", dict(kcopy_d2h=64, kcopy_h2d=64, kernels=False)),
                 ("H2D copy kernel only, no D2H", dict(kcopy_h2d=64, d2h=False, kernels=False)),
                 ("D2H copy kernel only, no H2D", dict(kcopy_d2h=64, h2d=False, kernels=False)),
                 ("full pipeline, one D2H copy per batch", {}),
                 ("full pipeline, three D2H copies", dict(split_d2h=True)),
                 ("copies only (no kernels)", dict(kernels=False)),
                 ("copies only, no events at all", dict(kernels=False, events=False)),
                 ("no H2D", dict(h2d=False)),
                 ("no D2H", dict(d2h=False)),
                 ("kernels only", dict(h2d=False, d2h=False))):
    run(3, **kw)
    print(f"{name:45s} {run(15, **kw):9.0f} frames/s", flush=True)

# the bench.py section itself in THIS process (same streams policy, its own context): does history matter?
r = bench.end_to_end_section(torch, api, dev, H, W, 0.15, Sw, Ux, Vxt, idx)
print(f"{'bench.end_to_end_section in this process':45s} {r['value']:9.0f} frames/s  (short runs of the stream sets tried: {r['stream_sets_tried_short_run_frames_per_s']})", flush=True)
