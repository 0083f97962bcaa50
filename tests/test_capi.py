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
