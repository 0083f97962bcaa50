"""Host-side duration of every enqueue call of the three-stream pipeline: which call blocks the issuing thread?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
F, H, W, nbuf = 8, 2160, 3840, 3
n = H * W; nt = n // 64
s_up, s_k, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
h_in = [torch.randint(0, 256, (F, H, W), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
ob = F * (2 * n + nt * 32)
h_out = [torch.empty(ob, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
d_in = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
d_out = [torch.empty(ob, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
x = torch.empty(64 << 20, dtype=torch.float32, device=dev)
def once(fresh_events, verbose=False):
    global ev_up, ev_k
    if fresh_events:
        ev_up = [torch.cuda.Event() for _ in range(nbuf)]; ev_k = [torch.cuda.Event() for _ in range(nbuf)]
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    for b in range(9):
        k = b % nbuf
        with torch.cuda.stream(s_up):
            d_in[k].copy_(h_in[k], non_blocking=True); ev_up[k].record(s_up)
        with torch.cuda.stream(s_k):
            s_k.wait_event(ev_up[k]); x.mul_(1.0001); ev_k[k].record(s_k)
        with torch.cuda.stream(s_dn):
            s_dn.wait_event(ev_k[k])
            h_out[k].copy_(d_out[k], non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t00) * 1e3 / 9


ev_up = [torch.cuda.Event() for _ in range(nbuf)]; ev_k = [torch.cuda.Event() for _ in range(nbuf)]
once(False)
print("reused events  :", " ".join(f"{once(False):.2f}" for _ in range(8)), "ms per batch")
print("fresh events   :", " ".join(f"{once(True):.2f}" for _ in range(8)), "ms per batch")
print("reused again   :", " ".join(f"{once(False):.2f}" for _ in range(4)), "ms per batch")
