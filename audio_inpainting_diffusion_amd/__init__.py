"""Import alias for the hyphenated package directory ``audio-inpainting-diffusion_amd/``.

``import audio_inpainting_diffusion_amd.network`` loads ``audio-inpainting-diffusion_amd/network.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "audio-inpainting-diffusion_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
