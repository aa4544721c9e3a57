#!/usr/bin/env python3
"""Derived golden vectors of the 5 nm K system, generated with the CPU oracle (itself pinned on the
reference's snapshot_6.xyz, see tests/test_oracle_golden.py).  Output: tests/golden/k5nm_golden.json.

    python tests/golden/make_golden_k5nm.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kmcf_oracle as O  # noqa: E402
import kmcfield_amd as km  # noqa: E402


def main():
    d = km.structure.load_device_5nm("init")
    NL = d["N_contact"]
    x, y, z = d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2]
    ks = O.KSystem(d["xyz"], d["lattice"], d["pbc"], d["nn_dist"], NL, NL)
    nl = O.neighbor_list(x, y, z, d["nn_dist"], 52)
    ch = O.update_charge(d["element"], np.zeros(d["N"], np.int32), nl, d["metals"])
    A = O.assemble_K(ks, d["element"], ch, d["metals"], d["high_G"], d["low_G"], d["Vd"])
    tol = 1e-14 * ks.n
    xs, it, rel = O.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], tol, 10000)
    x40, it40, rel40 = O.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], tol, 40)
    deg = np.diff(ks.row_ptr)
    sample = np.arange(0, ks.n, 97)
    halo = {}
    for P in (2, 4, 8):
        halo[str(P)] = [[int(h["rank"]) for h in O.halo_lists(ks.row_ptr, ks.col, P, r)] for r in range(P)]
        halo[str(P) + "_halo_cols"] = [int(sum(len(h["cols"]) for h in O.halo_lists(ks.row_ptr, ks.col, P, r)[1:]))
                                       for r in range(P)]
    out = dict(
        N=int(d["N"]), N_interface=int(ks.n), nnz=int(ks.nnz), left_nnz=int(len(ks.left_col)),
        right_nnz=int(len(ks.right_col)), degree_min=int(deg.min()), degree_max=int(deg.max()),
        degree_hist=np.bincount(deg).tolist(), col_checksum=int(ks.col.astype(np.int64).sum()),
        neigh_max=int((nl >= 0).sum(1).max()), neigh_checksum=int(nl.astype(np.int64).sum()),
        vacancies=int((d["element"] == 2).sum()), charged=int((ch != 0).sum()),
        charge_checksum=int((ch.astype(np.int64) * np.arange(d["N"])).sum()),
        diag_min=float(A["diag"].min()), diag_max=float(A["diag"].max()), rhs_nonzero=int((A["rhs"] != 0).sum()),
        spmv_ones_abs_sum=float(np.abs(O.spmv(ks.row_ptr, ks.col, A["val"], np.ones(ks.n))).sum()),
        tol=tol, pcg_iterations=int(it), pcg_relres=float(rel), x_min=float(xs.min()), x_max=float(xs.max()),
        x40_sample_idx=sample.tolist(), x40_sample=x40[sample].tolist(), pcg40_relres=float(rel40),
        halo=halo,
    )
    path = os.path.join(HERE, "k5nm_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, {k: out[k] for k in ("N_interface", "nnz", "charged", "pcg_iterations", "pcg_relres")})


if __name__ == "__main__":
    main()
