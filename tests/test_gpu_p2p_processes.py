"""GPU: the peer-to-peer transport between PROCESSES -- one process per rank as in production, real hipIpc memory
handles exchanged by the host program (torch.distributed gloo), no RCCL.  Two (and three) ranks share the box's
one GPU, which RCCL would refuse; the kernels of the ranks put into each other's windows, raise and poll sequence
flags and add the partial dot products in rank order exactly as they would across xGMI.  Checked against the
oracle's P-rank emulation: charges bit-exact, the solve converged to the reference's tolerance with the iteration
count of the single-reduction recurrence, every rank holds the same replicated solution, bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
from conftest import TRUE_RESIDUAL_BAR
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("P", [2, 3])
def test_ranks_as_processes_over_ipc_windows(km, oracle, dev5, ref5, tmp_path, P):
    port = _free_port()
    procs = []
    for r in range(P):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(P), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", KMCF_P2P_TIMEOUT_MS="20000",
                   KMCF_DEVICE_SHARE=str(P))      # (the P processes share the box's one GPU: grids that must be resident together take 1/P of it)
        env.pop("KMCF_TRANSPORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank process did not finish")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    d = dev5
    NL = d["N_contact"]
    ks, A = ref5["ks"], ref5["A"]
    xo, ito, relo = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], ref5["tol"], 10000, P=P)
    v0 = np.load(tmp_path / "v_0.npy")
    for r in range(P):
        st = json.load(open(tmp_path / ("st_%d.json" % r)))
        assert st["transport"] == "p2p"
        assert np.array_equal(np.load(tmp_path / ("charge_%d.npy" % r)), ref5["charge"])
        v = np.load(tmp_path / ("v_%d.npy" % r))
        assert np.array_equal(v, v0)                                         # replicated bit for bit
        assert st["st"]["converged"] == 1 and st["st"]["relres"] <= ref5["tol"]
        assert abs(st["st"]["iterations"] - ito) <= 0.05 * ito               # single-reduction recurrence (see test_gpu_multirank)
        print("rank %d of %d (processes, one GPU): %s" % (r, P, st["diag"]))
    dx = np.abs(v0[NL:-NL] - xo)
    assert dx.max() <= 5e-4 and np.median(dx) <= 5e-6
    res = A["rhs"] - oracle.spmv(ks.row_ptr, ks.col, A["val"], v0[NL:-NL])
    assert np.linalg.norm(res) / np.linalg.norm(A["rhs"]) <= TRUE_RESIDUAL_BAR
