"""GPU: the multi-rank code path of the CG (1-block finalize kernels + RCCL all-reduce of the dot
products on device scalars, RCCL gathers) exercised on ONE GPU: KMCF_FORCE_COMM=1 makes a 1-rank
group create its two RCCL communicators and run every collective.  (RCCL refuses two ranks on one
GPU, so N>1 proper cannot run on a 1-GPU box; the halo protocol is covered by tests/test_dist_gloo.py
and tests/test_abi.py, the 8-GPU run by the driver's scaling bench.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np, torch
import kmcfield_amd as km
S = km.solvers
d = km.structure.load_device_5nm("init")
NL = d["N_contact"]
comm = S.KMC_comm(d["N"] - 2 * NL, d["N"] + 1, d["N"], d["N"], rank=0, size=1, device=0)
comm.connect()
buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                   d["lattice"], d["metals"])
S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                    buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"],
                                       d["nn_dist"], len(d["metals"]), 0)
S.sum_and_gather_potential(buf, NL, comm)
v = buf.site_potential_boundary.cpu().numpy()
mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
tb = [mat.comm_bench(kind, 20) for kind in (0, 1, 2)]        # bench.py's N>1 diagnostic
print("RESULT " + json.dumps(dict(st=st, vsum=float(np.abs(v).sum()), charged=int((buf.site_charge != 0).sum().item()),
                                  tb=tb)))
""" % ROOT


def _run(force):
    env = dict(os.environ)
    env.pop("KMCF_FORCE_COMM", None)
    env["KMCF_CG_RESIDENT"] = "0"      # both runs as the loop of kernels (the plain one would otherwise be ONE resident launch, which adds in another order)
    if force:
        env["KMCF_FORCE_COMM"] = "1"
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_forced_collectives_match_plain_path():
    plain = _run(False)
    forced = _run(True)
    assert forced["st"]["converged"] == 1 and plain["st"]["converged"] == 1
    # a 1-rank all-reduce is the identity: same partial sums, same order -> identical iterates
    assert forced["st"]["iterations"] == plain["st"]["iterations"]
    assert forced["st"]["relres"] == pytest.approx(plain["st"]["relres"], rel=1e-12)
    assert forced["vsum"] == pytest.approx(plain["vsum"], rel=1e-12)
    assert forced["charged"] == plain["charged"] == 339
    assert all(t >= 0.0 for t in forced["tb"]) and forced["tb"][2] > 0.0
