"""Importable alias of the package whose sources live in ``neural-spectral-codec_amd/``.

The repo layout names the package directory with a hyphen, which Python cannot import; this
shim points the package search path at that directory and runs its ``__init__.py`` here, so
``import neural_spectral_codec_amd.encoding.spectral_encoder`` resolves to
``neural-spectral-codec_amd/encoding/spectral_encoder.py``.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "neural-spectral-codec_amd")
__path__[:] = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
