#!/usr/bin/env python3
"""Probe: does the device-order oracle reproduce the GPU's PCG bit for bit?  (5 nm device, single rank)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import kmcfield_amd as km
import kmcf_oracle as O
S = km.solvers
d = km.structure.load_device_5nm("init")
NL = d["N_contact"]
comm = S.KMC_comm(d["N"] - 2 * NL, d["N"] + 1, d["N"], d["N"], rank=0, size=1, device=0); comm.connect()
buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"], d["lattice"], d["metals"])
S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types, buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
kv = S.k_vectors(buf)
plan = mat.sum_plan()
print({k: v for k, v in plan.items() if not hasattr(v, "shape")})
n = plan["rows"]; tol = 1e-14 * n
rng = np.random.default_rng(1)
xv = rng.standard_normal(n)
p = torch.as_tensor(xv, device="cuda"); Ap = torch.empty_like(p); mat.spmv(p, Ap)
y = O.spmv_device_order(plan, xv)
print("spmv bitwise equal:", np.array_equal(y, Ap.cpu().numpy()), np.abs(y - Ap.cpu().numpy()).max())
for fixed in (1, 2, 5, 40, 0):
    r = torch.as_tensor(kv["rhs"], device="cuda").clone(); x = torch.zeros_like(r); dinv = torch.as_tensor(kv["dinv"], device="cuda")
    st = S.conjugate_gradient_jacobi(mat, r, x, dinv, tol, 10000, fixed_iters=fixed)
    o = O.pcg_device_order(plan, kv["rhs"], np.zeros(n), kv["dinv"], tol, 10000, fixed_iters=fixed)
    xg = x.cpu().numpy(); rg = r.cpu().numpy()
    print("fixed", fixed, "iters", st["iterations"], o["iterations"], "rz", st["rz"], o["rz"], "bb eq", st["bb"] == o["bb"],
          "x eq", np.array_equal(xg, o["x"]), "r eq", np.array_equal(rg, o["r"]), "max|dx|", np.abs(xg - o["x"]).max())
