import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def hostapi():
    return importlib.import_module(PKG_NAME + ".hostapi")


@pytest.fixture(scope="session")
def gpu_ctx(hostapi):
    """One wm_ctx on device 0 for the whole GPU session.  Fails loudly (no skip)
    when the HIP library or the GPU is missing: -m gpu tests must exercise the
    native path."""
    assert hostapi.device_count() >= 1, "no HIP device visible"
    ctx = hostapi.Context(0)
    yield ctx
    ctx.close()
