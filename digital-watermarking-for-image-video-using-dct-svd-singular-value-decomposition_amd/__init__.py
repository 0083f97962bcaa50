"""MI355X-native (gfx950) DCT-SVD watermark hot path.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/wmhip.h) -> libwmhip.so
  hostapi    ctypes binding of that ABI (no CPU fallback)
The directory name is not a Python identifier; import it with
``importlib.import_module(PACKAGE_NAME)`` (see ``dct_svd_core_secure.py`` at the
repository root, which is the drop-in module the reference's apps import).
"""
PACKAGE_NAME = __name__
__version__ = "0.1.0"
