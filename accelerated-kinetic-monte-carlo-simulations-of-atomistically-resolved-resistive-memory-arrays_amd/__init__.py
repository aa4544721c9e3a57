"""kmcfield: MI355X-native field solve (K-matrix assembly + distributed Jacobi-PCG) for DeviceKMC.

Layout: csrc/ holds the hand-written HIP kernels and the C ABI (include/kmcfield.h);
lib.py binds it with ctypes; solvers.py mirrors the reference's gpu_solvers.h /
dist_iterative call surface; structure.py provides the 5 nm fixture and the
synthetic 40 nm crossbar generator.
"""
from . import lib  # noqa: F401


def build(verbose=False):
    """Compile libkmcfield.so for gfx950 (hipcc cross-compiles without a GPU)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = ["make", "-C", os.path.join(here, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return lib.LIB_PATH
