"""Loader for the `robo-vln_amd/` package directory.

The directory name contains a hyphen (it is fixed by the project layout), so it
cannot be imported by name.  `load()` registers it in `sys.modules` under the
importable alias `robo_vln_amd`; after that `import robo_vln_amd.policy` etc.
work normally.
"""
import importlib.util
import os
import sys

ALIAS = "robo_vln_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "robo-vln_amd")


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
