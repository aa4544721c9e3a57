import os
import sys

# In-process rank groups that drive the device-side peer-to-peer protocol (KMCF_TRANSPORT=p2p) wait for each
# other ON the GPU: their streams must not share a hardware queue.  Read by the HIP runtime when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    import kmcf_oracle
    kmcf_oracle.lib()
    return kmcf_oracle


@pytest.fixture(scope="session")
def km():
    import kmcfield_amd
    if not os.path.exists(kmcfield_amd.lib.LIB_PATH):
        kmcfield_amd.build()
    return kmcfield_amd


@pytest.fixture(scope="session")
def dev5(km):
    return km.structure.load_device_5nm("init")


@pytest.fixture(scope="session")
def ref5(oracle, dev5):
    """Oracle results on the 5 nm device (pattern, neighbour list, charges, K system, cold PCG)."""
    import numpy as np
    d = dev5
    NL = d["N_contact"]
    ks = oracle.KSystem(d["xyz"], d["lattice"], d["pbc"], d["nn_dist"], NL, NL)
    nl = oracle.neighbor_list(d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], d["nn_dist"], 52)
    charge = oracle.update_charge(d["element"], np.zeros(d["N"], np.int32), nl, d["metals"])
    A = oracle.assemble_K(ks, d["element"], charge, d["metals"], d["high_G"], d["low_G"], d["Vd"])
    tol = 1e-14 * ks.n
    x, it, rel = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], tol, 10000)
    return dict(ks=ks, neigh=nl, charge=charge, A=A, x=x, iters=it, relres=rel, tol=tol)


# Iteration count of the cold K solve on the 5 nm device (x0 = 0, tolerance 1e-14 * N): the reference's own run logs
# 327 (expected_output/output1_0.txt), the oracle (natural row order, sequential sums) 317-318.
# The count moves by a few iterations with the order in which rows and partial sums are added (convergence is
# decided at 1e-14): two correct implementations differ by 3 %.  Gate: within 2 % (BASELINE.md's figure) of the
# interval those two span.
REFERENCE_ITERS_5NM = 327


def iters_in_gate(got, oracle_iters, rel=0.02):
    lo, hi = min(oracle_iters, REFERENCE_ITERS_5NM), max(oracle_iters, REFERENCE_ITERS_5NM)
    return lo * (1.0 - rel) <= got <= hi * (1.0 + rel)


# True residual ||b - A x|| / ||b|| of a converged K solve (5 nm device).  The loop stops on the RECURRENCE residual
# (sqrt(r.z / b.b) <= 1e-14 N); the true residual is whatever rounding has left between the two: 1.2e-9 for the oracle,
# 1.0e-9 ... 2.1e-9 observed on the GPU across row orders, rank counts and recurrences.
TRUE_RESIDUAL_BAR = 4e-9
