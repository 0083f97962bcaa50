"""The C-ABI library loads on a box without a GPU and exports every symbol
include/wmhip.h declares (no compute calls here)."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge


def _header_functions():
    txt = open(os.path.join(ge.ROOT, "include", "wmhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", txt)))


def test_header_binding_and_library_agree(hostapi):
    ge.build_hip()
    names = _header_functions()
    assert len(names) >= 25
    assert sorted(hostapi.SIGNATURES) == names            # binding types every declared entry point
    lib = hostapi.load_library()
    for n in names:
        assert hasattr(lib, n), f"libwmhip.so does not export {n}"
    assert lib.wm_abi_version() == hostapi.ABI_VERSION == 1


def test_no_gpu_fails_loudly_not_silently(hostapi):
    """Without a device the product raises; it never computes on the host."""
    if hostapi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises((hostapi.WmError, ValueError)):
        hostapi.Context(0)
    import dct_svd_core_secure as core
    with pytest.raises((hostapi.WmError, ValueError)):
        core.embed_arrays(np.zeros((16, 16, 3), np.uint8), np.zeros((4, 4, 3), np.uint8), "pw", bytes(8))


def test_missing_library_is_an_import_error(hostapi, tmp_path):
    with pytest.raises(hostapi.WmLibraryError):
        hostapi.load_library(str(tmp_path / "nope.so"))


def test_product_does_not_import_the_oracle():
    pkg_dir = os.path.join(ge.ROOT, ge.PKG_NAME)
    for root, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "wm_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
    assert "oracle" not in open(os.path.join(ge.ROOT, "dct_svd_core_secure.py")).read()


class _NoCallContext:
    """hostapi.Context without a device: any attempt to cross the C ABI fails the test."""

    def __new__(cls, hostapi):
        ctx = object.__new__(hostapi.Context)
        ctx._h = None

        def _call(name, *args):
            raise AssertionError(f"{name} reached the C ABI with unvalidated shapes")
        ctx._call = _call
        return ctx


def test_meta_shapes_are_validated_before_the_c_abi(hostapi):
    """a meta that does not belong to the stego (short Sc / Sw, wrong tile grid, per-plane
    factor batch of another size) must raise ValueError on the host - the C side would read
    min(H, W) or n_tiles * 8 floats from whatever buffer it is given."""
    ctx = _NoCallContext(hostapi)
    st = np.zeros((64, 96), np.uint8)
    good_sc = np.zeros(64, np.float32)
    for sc, sw in ((np.zeros(32, np.float32), good_sc), (good_sc, np.zeros(63, np.float32)),
                   (np.zeros((2, 64), np.float32), good_sc)):
        with pytest.raises(ValueError):
            ctx.ref_detect(st, sc, sw, 0.1)
    nby, nbx = 8, 12
    sc_t = np.zeros((nby, nbx, 8), np.float32)
    with pytest.raises(ValueError):
        ctx.detect_tiles(st, sc_t, np.zeros((nby, nbx - 1, 8), np.float32), 0.1)
    with pytest.raises(ValueError):
        ctx.detect_tiles(st, sc_t[:-1], sc_t, 0.1)
    with pytest.raises(ValueError):
        ctx.detect_tiles(st, sc_t, np.zeros((2, nby, nbx, 8), np.float32), 0.1)       # per-plane Sw for 1 plane
    U = np.zeros((nby, nbx, 8, 8), np.float32)
    with pytest.raises(ValueError):
        ctx.extract_tiles(st, sc_t, np.zeros((2, nby, nbx, 8, 8), np.float32),
                          np.zeros((2, nby, nbx, 8, 8), np.float32), 0.1)             # 5-D Uw with shape[0] != n
    with pytest.raises(ValueError):
        ctx.extract_tiles(st, sc_t[:4], U, U, 0.1)
    with pytest.raises(ValueError):
        ctx.extract_tiles(st, sc_t, U, U[:, :5], 0.1)
    with pytest.raises(ValueError):
        ctx.reconstruct_tiles(U, np.zeros((nby, nbx, 4), np.float32), U, 64, 96)
    # the scramble index is gathered / scattered through unchecked on the device: range-checked before the upload
    bad = np.arange(64 * 96); bad[7] = 64 * 96
    with pytest.raises(ValueError):
        ctx.permute_planes(st, bad)
    bad[7] = -1
    with pytest.raises(ValueError):
        ctx.permute_planes(st, bad)
    with pytest.raises(ValueError):
        ctx.permute_planes(st, np.arange(64 * 96 - 1))


def test_meta_tile_and_stego_size_are_checked():
    import dct_svd_core_secure as core
    mod = core.embed_arrays.__globals__
    meta = {"tile": np.int32(16), "Sc": np.zeros((2, 2, 8), np.float32), "shape": np.array((16, 16))}
    with pytest.raises(ValueError, match="tile must be 8 or None"):
        mod["_meta_tile"](meta)
    assert mod["_meta_tile"]({"tile": np.int32(0), "Sc": np.zeros(4)}) is None
    assert mod["_meta_tile"]({"tile": np.int32(8), "Sc": np.zeros(4)}) == 8
    assert mod["_meta_tile"]({"Sc": np.zeros((2, 2, 8))}) == 8 and mod["_meta_tile"]({"Sb": np.zeros(16)}) is None
    with pytest.raises(ValueError, match="meta was written for"):
        mod["_check_stego_shape"](np.zeros((16, 24, 3), np.uint8), meta)
    mod["_check_stego_shape"](np.zeros((16, 16, 3), np.uint8), meta)


class _FakeLib:
    """Stands in for libwmhip.so on a box without a GPU: every entry point succeeds and writes nothing, except
    wm_check_status, which reports the sticky status bit (set -> WM_ERR_NOCONV).  Records the calls."""

    def __init__(self, status_rc=0):
        self.status_rc = status_rc
        self.calls = []
        self._next = 0x1000

    def wm_last_error(self):
        return b"SVD did not converge"

    def __getattr__(self, name):
        if not name.startswith("wm_"):
            raise AttributeError(name)

        def fn(h, *args):
            self.calls.append(name)
            if name == "wm_check_status":
                return self.status_rc
            if name == "wm_malloc":
                self._next += 0x100000
                args[1]._obj.value = self._next
            return 0
        return fn


def _fake_ctx(hostapi, status_rc=0):
    ctx = object.__new__(hostapi.Context)
    ctx.lib = _FakeLib(status_rc)
    ctx._h = None                       # close() / __del__ do nothing
    ctx.__dict__["_h"] = 1
    ctx.device = 0
    return ctx


def test_device_resident_extract_reports_a_set_status_bit(hostapi):
    """extract_tiles_unscrambled_u8 drives the *_dev entry points, which only SET the sticky status bit: the wrapper
    must end in wm_check_status so that a Jacobi that hit its sweep bound raises LinAlgError here, not in
    some later unrelated call (round-2 advisor finding)."""
    H = W = 16
    st = np.zeros((H, W), np.uint8); sc = np.zeros((2, 2, 8), np.float32)
    U = np.zeros((2, 2, 8, 8), np.float32)
    idx = np.arange(H * W)
    ctx = _fake_ctx(hostapi, status_rc=hostapi.WM_ERR_NOCONV)
    with pytest.raises(np.linalg.LinAlgError):
        ctx.extract_tiles_unscrambled_u8(st, sc, U, U, 0.1, 8, idx)
    calls = ctx.lib.calls
    assert "wm_check_status" in calls and calls.index("wm_check_status") > calls.index("wm_extract_tiles_u8_dev")
    assert calls.count("wm_free") == calls.count("wm_malloc") - 1          # everything but the cached index is released
    ok = _fake_ctx(hostapi, status_rc=0)
    out = ok.extract_tiles_unscrambled_u8(st, sc, U, U, 0.1, 8, idx)
    assert out.shape == (H, W) and out.dtype == np.uint8
    ctx.__dict__["_h"] = None; ok.__dict__["_h"] = None


def test_device_index_cache_is_keyed_by_identity_not_address(hostapi):
    """The device copy of a permutation is cached on what the index IS - (H, W, sha256(key)) for
    hostglue.permutation_index, a content digest otherwise - never on its address (NumPy reuses a freed index's
    address for the next arange of the same size).  Non-bijective indices are refused before the upload."""
    import importlib
    hg = importlib.import_module(ge.PKG_NAME + ".hostglue")
    ctx = _fake_ctx(hostapi)
    a = hg.permutation_index(8, 8, b"k" * 32)
    assert a.tag == ("perm", 8, 8, __import__("hashlib").sha256(b"k" * 32).digest()) and a[::2].tag is None
    d1 = ctx.index_dev(a)
    assert ctx.index_dev(a) == d1 and ctx.lib.calls.count("wm_malloc") == 1
    b = hg.permutation_index(8, 8, b"q" * 32)
    assert ctx.index_dev(b) != d1
    # same contents at a different address: one device copy; different contents in the SAME buffer: another
    x = np.arange(64)[::-1].copy()
    dx = ctx.index_dev(x)
    assert ctx.index_dev(x.copy()) == dx
    x[0], x[1] = x[1], x[0]
    assert ctx.index_dev(x) != dx
    with pytest.raises(ValueError):
        ctx.index_dev(np.zeros(16, np.int64))                   # in range, but not a permutation
    with pytest.raises(ValueError):
        ctx.index_dev(np.arange(16) + 1)
    ctx.__dict__["_h"] = None


def test_extract_checks_the_hmac_before_anything_else_counts():
    """The HMAC over the meta's factors runs on a worker thread under the device work; the reference checks it first
    (single:206-209), so a mismatch must win over whatever else went wrong, and the other error must surface when the
    password is right.  No device needed: a tile-mode meta for another stego size fails before any device call."""
    import importlib
    import dct_svd_core_secure as core
    hg = importlib.import_module(core._impl.__package__ + ".hostglue")
    rng = np.random.default_rng(3)
    nonce = bytes(range(8))
    Sc = rng.normal(0, 1, (2, 2, 8)).astype(np.float32)
    Uw = rng.normal(0, 1, (2, 2, 8, 8)).astype(np.float32); Vwt = rng.normal(0, 1, (2, 2, 8, 8)).astype(np.float32)
    digest = hg.hmac_digest(hg.derive_key("right", nonce), [Sc, Uw, Vwt])
    meta = dict(mode="gray", alpha=0.1, shape=np.array((16, 16)), nonce=np.frombuffer(nonce, dtype=np.uint8),
                digest=np.frombuffer(digest, dtype=np.uint8), Sc=Sc, Uw=Uw, Vwt=Vwt, tile=np.int32(8), kfrac=0.6)
    stego_other_size = np.zeros((16, 24, 3), np.uint8)
    with pytest.raises(ValueError, match="Sai mật khẩu"):
        core.extract_arrays(stego_other_size, meta, "wrong")
    with pytest.raises(ValueError, match="meta was written for"):
        core.extract_arrays(stego_other_size, meta, "right")
