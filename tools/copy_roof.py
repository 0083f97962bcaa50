"""Practical HBM roof on the box: device-to-device copy bandwidth (read + write bytes
over HIP-event time) at a few sizes, for the 'achieved-copy GB/s' row of DESIGN.md."""
import json
import torch

dev = torch.device("cuda:0")
out = []
for mb in (64, 256, 1024, 4096):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 256)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out.append({"MiB": mb, "us": ms * 1e3, "GBps_read_plus_write": 2 * n / ms / 1e6})
    del a, b
print(json.dumps(out))
