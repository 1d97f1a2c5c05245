"""Importable alias of the ``physics-based-climate-model_amd`` package (its directory name is not a valid Python
identifier).  ``import climate_amd`` == that package; submodules resolve into its directory."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "physics-based-climate-model_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
