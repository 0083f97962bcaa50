"""``from dct_svd_core_secure import embed, extract, detect`` - the module name
the reference's apps import (app_dct_svd_pyside6.py:8).  Thin shim: the
implementation is the package module of the same name (its directory name is
not a Python identifier, hence importlib)."""
import importlib as _importlib

_impl = _importlib.import_module(
    "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd."
    "dct_svd_core_secure")

embed = _impl.embed
extract = _impl.extract
detect = _impl.detect
embed_watermark = _impl.embed_watermark
extract_watermark = _impl.extract_watermark
embed_arrays = _impl.embed_arrays
extract_arrays = _impl.extract_arrays
detect_arrays = _impl.detect_arrays
K_FRAC_DEFAULT = _impl.K_FRAC_DEFAULT

__all__ = ["embed", "extract", "detect", "embed_watermark", "extract_watermark",
           "embed_arrays", "extract_arrays", "detect_arrays", "K_FRAC_DEFAULT"]
