"""Import shim: the package directory name contains hyphens (it mirrors the reference
repository's name), so it cannot be imported with a plain `import`.  This module loads
it under the name `kmcfield_amd_pkg` and re-exports its submodules:

    import kmcfield_amd as km
    km.solvers.background_potential_gpu_sparse(...)
"""
import importlib.util
import os
import sys

PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "accelerated-kinetic-monte-carlo-simulations-of-atomistically-resolved-resistive-memory-arrays_amd")
_NAME = "kmcfield_amd_pkg"

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                   submodule_search_locations=[PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
pkg = sys.modules[_NAME]
build = pkg.build
lib = pkg.lib


def __getattr__(name):
    if name in ("solvers", "structure"):
        return importlib.import_module(_NAME + "." + name)
    raise AttributeError(name)
