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


def test_current_solve_between_processes_tiles_dealt_to_the_ranks(km, dev5, ref5, tmp_path, monkeypatch):
    """The T path between two rank PROCESSES over hipIpc windows with the tunnel block as dense symmetric tiles dealt to the
    ranks (kmcf_subop::spread): the sub-vector all-gather, the all-gather of the ranks' partial sums and the halo exchange
    all run through the peer-to-peer windows of separate processes.  Held against the same group as host threads of this
    process (whose summation order tests/test_gpu_tpath.py holds against the oracle): the transport does not enter the
    arithmetic -- iteration count, current and every potential identical, bit for bit."""
    import threading
    import torch
    import p2p_worker
    P = 2
    port = _free_port()
    procs = []
    for r in range(P):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(P), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   KMCF_P2P_TIMEOUT_MS="20000", KMCF_DEVICE_SHARE=str(P), KMCF_WORKER_T="1", KMCF_SUB_DENSE="1")
        env.pop("KMCF_TRANSPORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank process did not finish")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    # the same group as threads of this process
    S = km.solvers
    monkeypatch.setenv("KMCF_TRANSPORT", "p2p")
    monkeypatch.setenv("KMCF_P2P_TIMEOUT_MS", "20000")
    monkeypatch.setenv("KMCF_SUB_DENSE", "1")
    d = dev5
    N = d["N"]
    comms = S.KMC_comm.loopback_group(N - 2 * d["N_contact"], N + 1, N, N, P)
    out, errs = [None] * P, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            out[r] = p2p_worker.run_T(km, torch, comms[r], d, ref5["charge"])
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append("rank %d: %s\n%s" % (r, e, traceback.format_exc()))

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(240)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out), "a rank did not finish"
    for c in comms:
        c.close()
    for r in range(P):
        st = json.load(open(tmp_path / ("t_st_%d.json" % r)))
        assert st["info"]["tunnel_dense"] == 1 and st["info"]["tunnel_points"] == 1913 and st["st"]["converged"] == 1
        assert st["info"]["tunnel_bytes"] == out[r]["info"]["tunnel_bytes"] > 0
        assert st["st"]["iterations"] == out[r]["st"]["iterations"] and st["im"] == out[r]["im"]
        np.testing.assert_array_equal(np.load(tmp_path / ("t_v_%d.npy" % r)), out[0]["v"])
        np.testing.assert_array_equal(np.load(tmp_path / ("t_pw_%d.npy" % r)), out[0]["pw"])
    print("T 5 nm between %d processes, tiles dealt to the ranks: %d iterations, I_macro %.9e" % (P, out[0]["st"]["iterations"], out[0]["im"]))
