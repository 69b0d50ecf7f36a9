"""Importable alias of the package `2022_cambroise_interpret_multivae_amd`
(whose name is not a Python identifier): `import mopoe_amd as mm`."""
import importlib as _importlib
import sys as _sys

_pkg = _importlib.import_module("2022_cambroise_interpret_multivae_amd")
_sys.modules[__name__] = _pkg
