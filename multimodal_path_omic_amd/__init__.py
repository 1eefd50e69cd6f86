"""Importable alias of the package directory `multimodal-path-omic_amd/`.

A hyphen cannot appear in a Python module name, so `import multimodal_path_omic_amd`
resolves here and this stub re-points the package at the real directory: sub-module
imports (`multimodal_path_omic_amd.blocks`, ...) are served from there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "multimodal-path-omic_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
