"""GPU: the boundary driven from C++.  examples/kmc_loop.cpp -- a plain C++ / HIP program, no Python, no torch --
is compiled against include/kmcfield.h, linked with libkmcfield.so and run on the reference's 5 nm device
(tests/golden/device_5nm.bin).  Its six "KMC time is:" lines are compared with the reference's own run
(structures/5nm_device/expected_output/output1_0.txt, in the fixture as kmc_times): the same end-to-end pin as
tests/test_gpu_reference_trajectory.py, through the host language the reference is written in."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def exe(km, tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    pkg = os.path.dirname(km.lib.LIB_PATH)
    out = str(tmp_path_factory.mktemp("cpp") / "kmc_loop")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "kmc_loop.cpp"), "-L" + pkg, "-lkmcfield",
                           "-Wl,-rpath," + pkg, "-o", out])
    return out


def _run(exe, *flags):
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "device_5nm.bin"), *flags], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    times = [float(m) for m in re.findall(r"KMC time is: (\S+)", r.stdout)]
    steps = int(re.search(r"steps: (\d+)", r.stdout).group(1))
    return r.stdout, times, steps


def test_cpp_driver_reproduces_reference_kmc_times(exe, dev5):
    out, times, steps = _run(exe)
    assert steps == 6 and len(times) == 6, out                                  # the loop ends when kmc_time >= t_switch
    np.testing.assert_allclose(times, dev5["kmc_times"], rtol=2e-3)
    events = [int(m) for m in re.findall(r"Number of KMC events: (\d+)", out)]
    assert sum(events) == 8


def test_cpp_driver_with_current_solver(exe, dev5):
    """solve_current = 1 (the shipped parameters.txt asks for it; the reference's main skips it because comm_T is
    forced off, src/KMC_comm.h:243): CB edge, T assembly and the split PCG run every step; with heating off the
    events and KMC times are unchanged."""
    out, times, steps = _run(exe, "--current")
    assert steps == 6
    np.testing.assert_allclose(times, dev5["kmc_times"], rtol=2e-3)
    its = [int(m) for m in re.findall(r"iteration \(T\) = (\d+)", out)]
    res = [float(m) for m in re.findall(r"iteration \(T\) = \d+, relative residual = (\S+)", out)]
    assert len(its) == 6 and max(res) <= 1e-15 * 25681 and its[0] > 100
    # no warm-start gain from step to step: like the reference, the potentials are left scaled by G0 (7.7e-10) in
    # atom_virtual_potentials (src/current_solver_gpu.cu:2038-2040), which as a start vector is as good as zero
    assert max(its) <= 1.2 * min(its)
