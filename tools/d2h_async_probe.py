import torch, time, ctypes
dev = torch.device("cuda", 0)
n = 66355200
d = torch.empty(n, dtype=torch.uint8, device=dev); h = torch.empty(n, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
for rep in range(3):
    with torch.cuda.stream(s):
        t0 = time.perf_counter(); h.copy_(d, non_blocking=True); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"torch D2H copy_: call {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms")
for rep in range(3):
    with torch.cuda.stream(s):
        t0 = time.perf_counter(); d.copy_(h, non_blocking=True); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"torch H2D copy_: call {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
for rep in range(3):
    t0 = time.perf_counter(); rc = hip.hipMemcpyAsync(h.data_ptr(), d.data_ptr(), n, 2, s.cuda_stream); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"hipMemcpyAsync D2H rc={rc}: call {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms")
