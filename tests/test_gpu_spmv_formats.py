"""GPU: the SpMV kernels of libkmcfield against each other and against the oracle.

kmcf_spmv_plan picks one of three kernels per matrix (vec / stream / window) and, for the window kernel,
dictionary-codes the values when the off-diagonals take at most 62 distinct doubles (K and the CB-edge
system: -high_G, -low_G).  All of them must compute the same operator: stream and the un-coded window kernel
bit for bit (same products, same summation order), the coded window kernel up to the position of the
diagonal product in the row sum."""
import os
import ctypes as C

import numpy as np
from conftest import TRUE_RESIDUAL_BAR
import pytest

CODED_ON = int(os.environ.get("KMCF_SPMV_CODED", "1") != "0")   # the suite is green under KMCF_SPMV_CODED=0 too
SELL_ON = int(os.environ.get("KMCF_SPMV_SELL", "1") != "0")      # ... and under KMCF_SPMV_SELL=0
# info["spmv_coded"]: 0 values streamed, 1 coded window kernel, 2 coded row-per-lane kernel (<= 3 values, rows <= 64)
CODED_K = (2 if SELL_ON else 1) * CODED_ON

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _replan(km, mat, monkeypatch, **env):
    for k in ("KIND", "U", "WQ", "LPR", "LPR2", "CODED", "SELL", "SELL_ROWS", "SELLV"):
        monkeypatch.delenv("KMCF_SPMV_" + k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv("KMCF_SPMV_" + k, str(v))
    lib = km.lib.load()
    lib.kmcf_spmv_replan.argtypes = [C.c_void_p]
    km.lib.check(lib.kmcf_spmv_replan(mat.handle), "kmcf_spmv_replan")
    return mat.info()


def test_k_matrix_all_kernels_agree(km, oracle, dev5, ref5, torch, monkeypatch):
    S = km.solvers
    d = dev5
    NL = d["N_contact"]
    ks, A = ref5["ks"], ref5["A"]
    comm = S.KMC_comm(d["N"] - 2 * NL, d["N"] + 1, d["N"], d["N"])
    comm.connect()
    buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                        buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
    S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
    mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
    info = mat.info()
    # default plan of the (brick-ordered) K matrix: window kernel, values coded by the assembly
    assert info["spmv_kind"] == 2 and info["spmv_coded"] == CODED_K
    assert info["spmv_tiles"] > 0 and 0 < info["spmv_window_cols"] < info["nnz"]
    rng = np.random.default_rng(3)
    x = rng.standard_normal(ks.n)
    want = oracle.spmv(ks.row_ptr, ks.col, A["val"], x)
    bound = oracle.spmv(ks.row_ptr, ks.col, np.abs(A["val"]), np.abs(x))
    p = torch.as_tensor(x, device="cuda")

    def run():
        Ap = torch.empty_like(p)
        mat.spmv(p, Ap)
        return Ap.cpu().numpy()

    y_coded = run()
    assert np.all(np.abs(y_coded - want) <= 1e-13 * bound)
    # the assembled values are what the reference rule gives, coded or not
    assert np.abs(mat.get_values() - A["val"]).max() <= 1e-12 * np.abs(A["val"]).max()
    res = {}
    for name, env in (("window_plain", dict(KIND=2, CODED=0, SELLV=0)), ("lane_f64", dict(KIND=2, CODED=0)), ("stream", dict(KIND=1)), ("vec", dict(KIND=0)),
                      ("window_coded", dict(KIND=2, SELL=0)), ("window_u4", dict(KIND=2, U=4, WQ=2, SELL=0)),
                      ("window_wq4", dict(KIND=2, WQ=4, CODED=0, SELLV=0)), ("lane", dict(KIND=2, SELL=1)),
                      ("lane_rows128", dict(KIND=2, SELL=1, SELL_ROWS=128))):
        inf = _replan(km, mat, monkeypatch, **env)
        assert inf["spmv_kind"] == env["KIND"], (name, inf)
        want_coded = 0 if env["KIND"] != 2 or not env.get("CODED", 1) else (2 if env.get("SELL", 1) else 1)
        assert inf["spmv_coded"] == want_coded, (name, inf)
        res[name] = run()
        assert np.all(np.abs(res[name] - want) <= 1e-13 * bound), name
    np.testing.assert_array_equal(res["window_plain"], res["stream"])        # same products, same order
    # the row-per-lane kernel with f64 values (what a matrix without a value dictionary gets): the coded kernel's products
    # (the dictionary values ARE the values) in the coded kernel's order (stored order, diagonal last): identical
    np.testing.assert_array_equal(res["lane_f64"], res["lane"])
    np.testing.assert_array_equal(res["window_plain"], res["window_wq4"])
    assert np.all(np.abs(res["window_u4"] - res["window_coded"]) <= 1e-15 * bound + 1e-300)   # both coded: diagonal last
    # the row-per-lane kernel adds a row's products in column order, whatever its tiling
    np.testing.assert_array_equal(res["lane_rows128"], res["lane"])
    assert np.all(np.abs(res["window_coded"] - res["lane"]) <= 4e-16 * 53 * bound)
    if CODED_K == 2:
        np.testing.assert_array_equal(y_coded, res["lane"])
    assert np.all(np.abs(res["window_plain"] - y_coded) <= 4e-16 * 53 * bound)

    # the solve itself, coded against plain values: same system, rounding-level different iterates
    sols = {}
    for name, env in (("coded", dict()), ("plain", dict(CODED=0))):
        _replan(km, mat, monkeypatch, **env)
        S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
        assert mat.info()["spmv_coded"] == (2 if name == "coded" else 0)      # (_replan clears the session's overrides)
        buf.site_potential_boundary.zero_()        # the solve starts from the previous potential (warm start)
        st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"],
                                               d["nn_dist"], len(d["metals"]), 0)
        assert st["converged"] == 1
        sols[name] = (st["iterations"], buf.site_potential_boundary.cpu().numpy().copy())
    assert abs(sols["coded"][0] - sols["plain"][0]) <= 0.05 * sols["plain"][0]
    dv = np.abs(sols["coded"][1] - sols["plain"][1])
    assert dv.max() <= 5e-4 and np.median(dv) <= 5e-6
    for name in sols:
        v = sols[name][1][NL:-NL]
        r = A["rhs"] - oracle.spmv(ks.row_ptr, ks.col, A["val"], v)
        assert np.linalg.norm(r) / np.linalg.norm(A["rhs"]) <= TRUE_RESIDUAL_BAR
    buf.freeGPUmemory()
    comm.close()


def _banded(n, rng, values):
    """Symmetric banded pattern (window-friendly), off-diagonal values drawn from `values` (or all distinct
    if None), diagonally dominant."""
    import scipy.sparse as sp
    offs = [1, 2, 5, 40, 41, 300]
    diags = []
    for o in offs:
        m = n - o
        if values is None:
            v = -rng.random(m) - 0.01
        else:
            v = rng.choice(values, m)
        diags.append(v)
    M = sp.diags(diags, offs, shape=(n, n), format="csr")
    M = (M + M.T).tocsr()
    M = (M + sp.diags(np.abs(M).sum(1).A1 * (1.0 + rng.random(n)))).tocsr()
    M.sort_indices()
    return M


def test_generic_matrix_value_coding(km, torch):
    """create_csr / set_values: coded when the off-diagonals take few distinct values (any diagonal), plain
    window kernel otherwise; get_values always returns what was set."""
    S = km.solvers
    rng = np.random.default_rng(11)
    n = 20000
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    M3 = _banded(n, rng, np.array([-1.0, -1e-8, -0.25]))
    Mg = _banded(n, rng, None)
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.empty_like(p)
    mat = S.Distributed_matrix(comm, n, [n], [0], M3.indices, M3.indptr, M3.data)
    info = mat.info()
    assert info["spmv_kind"] == 2 and info["spmv_coded"] == CODED_K, info
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), M3 @ x, rtol=1e-13, atol=1e-13 * np.abs(M3 @ x).max())
    np.testing.assert_array_equal(mat.get_values(), M3.data)
    # same pattern, all-distinct values: falls back to the value stream, same object
    assert (Mg.indptr == M3.indptr).all() and (Mg.indices == M3.indices).all()
    mat.set_values(Mg.data)
    assert mat.info()["spmv_coded"] == 0
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), Mg @ x, rtol=1e-13, atol=1e-13 * np.abs(Mg @ x).max())
    np.testing.assert_array_equal(mat.get_values(), Mg.data)
    # and back; -0.0 and 0.0 are different dictionary entries (bit patterns)
    v = M3.data.copy()
    v[v == -0.25] = -0.0
    mat.set_values(v)
    assert mat.info()["spmv_coded"] == CODED_K
    mat.spmv(p, Ap)
    Mz = M3.copy()
    Mz.data = v
    np.testing.assert_allclose(Ap.cpu().numpy(), Mz @ x, rtol=1e-13, atol=1e-13 * np.abs(Mz @ x).max())
    # Jacobi-PCG on the coded matrix
    b = rng.standard_normal(n)
    r = torch.as_tensor(b.copy(), device="cuda")
    xs = torch.zeros(n, dtype=torch.float64, device="cuda")
    mat.set_values(M3.data)
    st = S.conjugate_gradient_jacobi(mat, r, xs, torch.as_tensor(1.0 / M3.diagonal(), device="cuda"), 1e-12, 5000)
    assert st["converged"] == 1
    assert np.abs(M3 @ xs.cpu().numpy() - b).max() <= 1e-9
    mat.close()
    comm.close()


def test_scattered_columns_decline_the_window(km, torch):
    """Rows whose columns are spread over the whole vector fill a tile's window after a handful of rows: the
    plan falls back to the stream kernel (direct gathers)."""
    import scipy.sparse as sp
    S = km.solvers
    rng = np.random.default_rng(12)
    n = 30000
    B = sp.random(n, n, density=12.0 / n, random_state=np.random.RandomState(3), format="csr")
    M = (B + B.T + sp.diags(np.full(n, 50.0))).tocsr()
    M.sort_indices()
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    assert mat.info()["spmv_kind"] == 1
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.empty_like(p)
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), M @ x, rtol=1e-12, atol=1e-12)
    mat.close()
    comm.close()
