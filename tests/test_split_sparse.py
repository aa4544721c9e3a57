"""The "split sparse" operator of the T-matrix path (A_neighbour + P^T A_sub P,
dist_iterative/dist_conjugate_gradient_split_sparse.cpp, dist_spmv_split_sparse.cpp): merged into the
row-partitioned CSR at build time.  The reference's own test validates its split variants against the
MONOLITHIC CSR run through the same CG (dist_iterative_test/main_test_cg_split.cpp:17-143, 1311-1317,
relative L2 error); same idea here, with scipy as the monolithic reference."""
import ctypes as C

import numpy as np
import pytest


def _system(seed=3, n=6000, ns=400):
    """Banded 'neighbour' matrix + a dense-ish symmetric sub-block on ns scattered 'tunnel' rows."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    offs = [-7, -3, -1, 1, 3, 7]
    A = sp.diags([-np.ones(n - abs(o)) for o in offs], offs, format="csr")
    sub_rows = np.sort(rng.choice(n, ns, replace=False)).astype(np.int32)
    B = sp.random(ns, ns, density=0.08, random_state=np.random.RandomState(seed), format="csr")
    B = -(B + B.T)
    B.setdiag(0)
    B.eliminate_zeros()
    B = B.tocsr()
    B.sort_indices()
    # diagonals: row sums -> diagonally dominant SPD, placed in the neighbour part and in the sub-block
    A = (A + sp.diags(-A.sum(1).A1 + 0.5)).tocsr()
    B = (B + sp.diags(-B.sum(1).A1)).tocsr()
    A.sort_indices()
    B.sort_indices()
    P = sp.csr_matrix((np.ones(ns), (np.arange(ns), sub_rows)), shape=(ns, n))
    M = (A + P.T @ B @ P).tocsr()
    return A, B, sub_rows, M


@pytest.mark.parametrize("P", [1, 2, 3])
def test_split_sparse_planning_matches_monolithic(km, oracle, P):
    """CPU (host-only communicators): the merged operator's halo plan = the plan of the monolithic CSR."""
    S = km.solvers
    A, B, sub_rows, M = _system()
    n, ns = A.shape[0], B.shape[0]
    counts, displs = oracle.partition(n, P)
    owner = np.searchsorted(displs, sub_rows, side="right") - 1
    count_sub = np.bincount(owner, minlength=P).astype(np.int32)
    displ_sub = np.concatenate([[0], np.cumsum(count_sub)[:-1]]).astype(np.int32)
    M.sort_indices()
    lib = km.lib.load()
    for r in range(P):
        h = C.c_void_p()
        km.lib.check(lib.kmcf_comm_create(C.byref(h), -1, P, r), "comm")

        class _Comm:
            handle = h
        r0, nr = int(displs[r]), int(counts[r])
        rp = (A.indptr[r0:r0 + nr + 1] - A.indptr[r0]).astype(np.int32)
        col = A.indices[A.indptr[r0]:A.indptr[r0 + nr]]
        val = A.data[A.indptr[r0]:A.indptr[r0 + nr]]
        s0, nsl = int(displ_sub[r]), int(count_sub[r])
        srp = (B.indptr[s0:s0 + nsl + 1] - B.indptr[s0]).astype(np.int32)
        scol = B.indices[B.indptr[s0]:B.indptr[s0 + nsl]]
        sval = B.data[B.indptr[s0]:B.indptr[s0 + nsl]]
        m = S.Distributed_matrix.split_sparse(_Comm, n, counts, displs, col, rp, val, ns, count_sub, displ_sub,
                                              sub_rows, srp, scol if len(scol) else np.zeros(1, np.int32),
                                              sval if len(sval) else np.zeros(1))
        want = oracle.halo_lists(M.indptr.astype(np.int32), M.indices.astype(np.int32), P, r)
        got = m.neighbours()
        assert [g["rank"] for g in got] == [w["rank"] for w in want]
        for g, w in zip(got[1:], want[1:]):
            assert np.array_equal(g["cols"], w["cols"]) and np.array_equal(g["rows"], w["rows"])
        m.close()
        lib.kmcf_comm_destroy(h)


@pytest.mark.gpu
def test_split_sparse_pcg_matches_monolithic(km):
    import scipy.sparse.linalg as spla
    import torch
    S = km.solvers
    A, B, sub_rows, M = _system()
    n, ns = A.shape[0], B.shape[0]
    rng = np.random.default_rng(8)
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    m = S.Distributed_matrix.split_sparse(comm, n, [n], [0], A.indices, A.indptr, A.data, ns, [ns], [0], sub_rows,
                                          B.indptr, B.indices, B.data)
    assert m.info()["nnz"] == A.nnz + B.nnz
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.empty_like(p)
    m.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), M @ x, rtol=1e-12, atol=1e-12)
    b = rng.standard_normal(n)
    # assemble_preconditioner (src/current_solver_gpu.cu:1323-1330): diagonal = neighbour diag + tunnel diag
    dinv = torch.as_tensor(1.0 / M.diagonal(), device="cuda")
    r = torch.as_tensor(b.copy(), device="cuda")
    xs = torch.zeros(n, dtype=torch.float64, device="cuda")
    st = S.conjugate_gradient_jacobi(m, r, xs, dinv, 1e-13, 5000)
    assert st["converged"] == 1
    xref = spla.spsolve(M.tocsc(), b)
    err = np.sqrt(((xs.cpu().numpy() - xref) ** 2).sum() / (xref ** 2).sum())   # main_test_cg_split.cpp:1311-1317
    assert err <= 1e-10
    m.close()
    comm.close()
