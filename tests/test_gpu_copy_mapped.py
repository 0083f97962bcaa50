"""wm_copy_mapped_dev: the copy kernel between device memory and MAPPED pinned host memory.  Only pointers a
kernel can dereference are accepted - ROCm reports ordinary pageable host memory as hipMemoryTypeUnregistered with
hipSuccess, and a kernel that touched it would fault the GPU instead of returning an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pageable_host_memory_is_refused(gpu_ctx):
    n = 4096
    d = gpu_ctx.malloc(n)
    try:
        a = np.arange(n, dtype=np.uint8)                     # pageable
        with pytest.raises(ValueError):
            gpu_ctx.copy_mapped(d, a.ctypes.data, n)
        with pytest.raises(ValueError):
            gpu_ctx.copy_mapped(a.ctypes.data, d, n)
        with pytest.raises(ValueError):
            gpu_ctx.copy_mapped(d, 0, n)
        gpu_ctx.copy_mapped(d, a.ctypes.data, 0)             # an empty copy touches nothing
        gpu_ctx.sync()
        gpu_ctx.check_status()
    finally:
        gpu_ctx.free(d)


class _Pinned:
    """hipHostMalloc'd (pinned, device-mapped) bytes as a NumPy array - through the HIP runtime directly, no torch."""
    def __init__(self, n):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        rc = self.hip.hipHostMalloc(C.byref(self.ptr), C.c_size_t(n), C.c_uint(0))
        assert rc == 0 and self.ptr.value, rc
        self.arr = np.frombuffer((C.c_uint8 * n).from_address(self.ptr.value), dtype=np.uint8)

    def close(self):
        self.arr = None
        self.hip.hipHostFree(self.ptr)


def test_copies_are_byte_exact_both_ways_for_every_alignment(gpu_ctx):
    n = (1 << 20) + 37
    hs, hb = _Pinned(n), _Pinned(n)
    src, back = hs.arr, hb.arr
    src[:] = np.random.default_rng(9).integers(0, 256, n, dtype=np.uint8)
    p_src, p_back = hs.ptr.value, hb.ptr.value
    d = gpu_ctx.malloc(n + 64)
    try:
        for off_d, off_h, size in ((0, 0, n), (0, 0, 16), (1, 0, 4097), (0, 3, 65537), (5, 7, 1000), (16, 16, n - 16), (0, 0, 15), (3, 3, 1)):
            gpu_ctx.memset(d, 0xEE, n + 64)
            gpu_ctx.copy_mapped(d + off_d, p_src + off_h, size, 16)          # pinned host -> device
            back[:] = 0
            gpu_ctx.copy_mapped(p_back + off_h, d + off_d, size, 0)           # device -> pinned host
            gpu_ctx.sync()
            assert np.array_equal(back[off_h:off_h + size], src[off_h:off_h + size]), (off_d, off_h, size)
            assert int(back[:off_h].sum()) == 0 and int(back[off_h + size:].sum()) == 0
        # device -> device works as well
        d2 = gpu_ctx.malloc(n)
        gpu_ctx.copy_mapped(d, p_src, n)
        gpu_ctx.copy_mapped(d2, d, n)
        out = np.empty(n, np.uint8)
        gpu_ctx.sync()
        gpu_ctx.d2h(out, d2)
        gpu_ctx.sync()
        assert np.array_equal(out, src)
        gpu_ctx.free(d2)
        # a copy that runs past the end of the device allocation is refused, not launched
        with pytest.raises(ValueError):
            gpu_ctx.copy_mapped(d, p_src, n + 65)
        gpu_ctx.check_status()
    finally:
        gpu_ctx.free(d)
        hs.close(); hb.close()
