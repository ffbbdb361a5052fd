"""Import shim: the package sources live in ``shot-vae_amd/`` (a directory name Python cannot
import directly).  ``import shot_vae_amd`` resolves here and continues in that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "shot-vae_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
