"""Hyphenated alias of the package ``audio_inpainting_diffusion_amd`` (the real, importable directory).

The reference resolves plugin classes with ``importlib.import_module`` (utils/dnnlib/util.py:235-297), which
also accepts the repository's own hyphenated name, e.g.
``network.callable=audio-inpainting-diffusion_amd.network.Unet_CQT_oct_with_attention``.  This alias registers
the SAME module objects under both names (one class identity, one loaded ``libaid_hip.so``): nothing is executed twice.
"""
import importlib as _importlib
import pkgutil as _pkgutil
import sys as _sys

_real = _importlib.import_module("audio_inpainting_diffusion_amd")
for _m in _pkgutil.iter_modules(_real.__path__):
    if _m.name.startswith("lib"):          # libaid_hip.so is a C-ABI library (ctypes), not a Python extension module
        continue
    _sys.modules[__name__ + "." + _m.name] = _importlib.import_module(_real.__name__ + "." + _m.name)
_sys.modules[__name__] = _real
