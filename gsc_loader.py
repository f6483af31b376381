"""Imports the package directory ``gnark-symmetric-crypto_amd`` (not a valid Python identifier) as module ``gsc_amd``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "gsc_amd" in sys.modules:
        return sys.modules["gsc_amd"]
    pkg = os.path.join(_ROOT, "gnark-symmetric-crypto_amd")
    spec = importlib.util.spec_from_file_location("gsc_amd", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gsc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
