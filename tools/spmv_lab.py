#!/usr/bin/env python3
"""Tuning aid (GPU box): time SpMV kernel variants of libkmcfield on the synthetic 40 nm K matrix.

    python tools/spmv_lab.py [--workload 40nm|small] "KIND=0,LPR=16" "KIND=1,U=8,LPR2=4" ...
Each spec sets KMCF_SPMV_<key> env vars, re-plans the matrix, checks the result against the first
variant and prints us/launch and GB/s on the algorithmic bytes (12 nnz + 20 n)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmcfield_amd as km  # noqa: E402


def main():
    args = sys.argv[1:]
    workload = "40nm"
    if args and args[0] == "--workload":
        workload = args[1]
        args = args[2:]
    specs = args or ["KIND=0,LPR=16", "KIND=1,U=8,LPR2=4"]
    S = km.solvers
    order = os.environ.get("LAB_ORDER", "bwmin")
    d = km.structure.synth_crossbar_40nm(order=order, tiles=int(os.environ.get("LAB_TILES", "8"))) if workload == "40nm" else km.structure.synth_small(tiles=2, order=order)
    print("order", order)
    NL = d["N_contact"]
    comm = S.KMC_comm(d["N"] - 2 * NL, d["N"] + 1, d["N"], d["N"])
    comm.connect()
    buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                        buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
    S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
    lib = km.lib.load()
    lib.kmcf_spmv_replan.argtypes = [C.c_void_p]
    mat = S.Distributed_matrix.from_handle(lib.kmcf_kstate_matrix(buf.K_distributed))
    info = mat.info()
    n, nnz = info["rows_this_rank"], info["nnz"]
    alg = 12.0 * nnz + 20.0 * n
    print("matrix: n=%d nnz=%d alg_bytes=%.1f MB" % (n, nnz, alg / 1e6), flush=True)
    rng = np.random.default_rng(0)
    p = torch.as_tensor(rng.standard_normal(n), device="cuda")
    ref = None
    for spec in specs:
        for k in list(os.environ):
            if k.startswith("KMCF_SPMV_"):
                del os.environ[k]
        for kv in spec.split(","):
            k, v = kv.split("=")
            os.environ["KMCF_SPMV_" + k] = v
        km.lib.check(lib.kmcf_spmv_replan(mat.handle), "replan")
        Ap = torch.empty_like(p)
        mat.spmv(p, Ap)
        y = Ap.cpu().numpy()
        if ref is None:
            ref = y
        err = np.abs(y - ref).max() / np.abs(ref).max()
        best = 1e30
        for dot in ((os.environ.get("LAB_DOT", "1") != "0"),):
            mat.spmv_bench(5, dot)
            for _ in range(3):
                best = min(best, mat.spmv_bench(30, dot) / 30 * 1e3)
        print("%-28s %8.2f us  %7.1f GB/s  frac %.3f  relerr %.1e" % (spec, best, alg / best / 1e3, alg / best / 1e3 / 8000, err),
              flush=True)


if __name__ == "__main__":
    main()
