"""GPU: edge cases of the path -- periodic boundary patterns, generic caller-supplied CSR matrices
(ragged rows, empty rows, long rows that fall back to the row-per-wave kernel), the authors' binary
matrix format, unpreconditioned CG, zero right-hand side."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _comm(km, n):
    S = km.solvers
    c = S.KMC_comm(n, n, n, n)
    c.connect()
    return c


def test_pattern_pbc_matches_oracle(km, oracle, torch):
    """pbc = 1: minimum image in y and z (src/gpu_solvers.h:290-309); the periodic cell grid on the GPU
    against the oracle's brute-force loop.  Random sites in a 60 x 21 x 21 A box (6 cells across y, z)."""
    S = km.solvers
    rng = np.random.default_rng(2)
    N, NL = 3000, 50
    L = np.array([60.0, 21.0, 21.0])
    xyz = rng.random((N, 3)) * L
    xyz = xyz[np.argsort(xyz[:, 0])]          # contacts = lowest / highest x
    comm = _comm(km, N - 2 * NL)
    buf = S.GPUBuffers(N, np.full(N, 3, np.int32), xyz[:, 0], xyz[:, 1], xyz[:, 2], 52, 3.5e-10, 1.0, L, [6, 8])
    S.initialize_sparsity_K(buf, 1, 3.5, NL, comm)
    n = N - 2 * NL
    want = [oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 1, 3.5, n, n, NL, NL, brute=True),
            oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 1, 3.5, n, NL, NL, 0, brute=True),
            oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 1, 3.5, n, NL, NL, NL + n, brute=True)]
    for which in range(3):
        rp, col = S.k_pattern(buf, which)
        assert np.array_equal(rp, want[which][0]) and np.array_equal(col, want[which][1]), which
    # more entries than without pbc: wrap-around neighbours exist
    rp0, _ = oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 0, 3.5, n, n, NL, NL)
    assert want[0][0][-1] > rp0[-1]
    buf.freeGPUmemory()
    comm.close()


def _random_spd(n, rng, long_row=False):
    import scipy.sparse as sp
    B = sp.random(n, n, density=4.0 / n, random_state=np.random.RandomState(5), format="csr")
    M = (B + B.T).tolil()
    if long_row:                      # one row far longer than a stream chunk -> row-per-wave kernel
        idx = rng.choice(n, 3000, replace=False)
        for j in idx:
            M[7, j] = M[j, 7] = 0.01
    M = M.tocsr()
    M = M + sp.diags(np.abs(M).sum(1).A1 + 1.0)
    M = M.tocsr()
    M.sort_indices()
    return M


@pytest.mark.parametrize("long_row", [False, True])
def test_generic_csr_spmv_and_cg(km, oracle, torch, long_row):
    """Distributed_matrix ctor 1 on an arbitrary symmetric matrix with ragged rows (some hold only the
    diagonal); SpMV against scipy, Jacobi-PCG and plain CG to 1e-12."""
    S = km.solvers
    rng = np.random.default_rng(9)
    n = 5000
    M = _random_spd(n, rng, long_row)
    comm = _comm(km, n)
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    assert mat.info()["nnz"] == M.nnz
    x = rng.standard_normal(n)
    p = torch.as_tensor(x, device="cuda")
    Ap = torch.empty_like(p)
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), M @ x, rtol=1e-12, atol=1e-12)
    b = rng.standard_normal(n)
    for precond in (True, False):
        r = torch.as_tensor(b.copy(), device="cuda")
        xs = torch.zeros(n, dtype=torch.float64, device="cuda")
        dinv = torch.as_tensor(1.0 / M.diagonal(), device="cuda") if precond else None
        st = S.conjugate_gradient_jacobi(mat, r, xs, dinv, 1e-12, 2000)
        assert st["converged"] == 1 and st["relres"] <= 1e-12
        assert np.abs(M @ xs.cpu().numpy() - b).max() <= 1e-9
        # r holds the final residual b - A x (in/out like the reference's r_local_d)
        np.testing.assert_allclose(r.cpu().numpy(), b - M @ xs.cpu().numpy(), atol=1e-9)
    # values can be replaced in place (creation order)
    mat.set_values(2.0 * M.data)
    np.testing.assert_allclose(mat.get_values(), 2.0 * M.data)
    mat.spmv(p, Ap)
    np.testing.assert_allclose(Ap.cpu().numpy(), 2.0 * (M @ x), rtol=1e-12, atol=1e-12)
    mat.close()
    comm.close()


def test_zero_rhs_and_converged_start(km, torch):
    """b = 0: r.z/b.b is NaN -> the loop never runs, like the reference's while condition (:217)."""
    S = km.solvers
    rng = np.random.default_rng(1)
    n = 2000
    M = _random_spd(n, rng)
    comm = _comm(km, n)
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    r = torch.zeros(n, dtype=torch.float64, device="cuda")
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    dinv = torch.as_tensor(1.0 / M.diagonal(), device="cuda")
    st = S.conjugate_gradient_jacobi(mat, r, x, dinv, 1e-10, 100)
    assert st["iterations"] == 0
    assert np.all(x.cpu().numpy() == 0)
    mat.close()
    comm.close()


def test_authors_binary_matrix_format(km, torch, tmp_path):
    """dist_iterative_test/utils.cpp:25-56 + main_test_cg.cpp:140-170: raw host-endian binaries
    A_data / A_row_ptr / A_col_indices / A_rhs (+ solution), tol 1e-11, zero start, and the authors'
    error measure sum|x - x_ref| / sum|x_ref| (main_test_cg.cpp:125-135)."""
    S = km.solvers
    rng = np.random.default_rng(4)
    n = 4000
    M = _random_spd(n, rng)
    xref = rng.standard_normal(n)
    b = M @ xref
    M.data.astype(np.float64).tofile(tmp_path / "A_data.bin")
    M.indptr.astype(np.int32).tofile(tmp_path / "A_row_ptr.bin")
    M.indices.astype(np.int32).tofile(tmp_path / "A_col_indices.bin")
    b.tofile(tmp_path / "A_rhs.bin")
    xref.tofile(tmp_path / "solution.bin")
    sysd = km.structure.load_csr_binaries(str(tmp_path), n, M.nnz)
    comm = _comm(km, n)
    counts, displs = S.KMC_comm.partition(n, 1)
    mat = S.Distributed_matrix(comm, n, counts, displs, sysd["col_indices"], sysd["row_ptr"], sysd["data"])
    r = torch.as_tensor(sysd["rhs"], device="cuda").clone()
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    diag = sysd["data"][[np.flatnonzero(sysd["col_indices"][sysd["row_ptr"][i]:sysd["row_ptr"][i + 1]] == i)[0]
                         + sysd["row_ptr"][i] for i in range(n)]]
    dinv = torch.as_tensor(1.0 / diag, device="cuda")
    st = S.conjugate_gradient_jacobi(mat, r, x, dinv, 1e-11, 10000)
    assert st["converged"] == 1
    xs = x.cpu().numpy()
    assert np.abs(xs - sysd["solution"]).sum() / np.abs(sysd["solution"]).sum() <= 1e-9
    mat.close()
    comm.close()


def test_entry_points_wait_for_the_callers_queued_work(km, torch):
    """Stream ordering contract (include/kmcfield.h): the library's non-blocking streams are ordered after
    the work the caller has queued on its own stream.  Here ~40 ms of small kernels that build the input
    vector are still in flight on torch's default stream when kmcf_spmv is called; without the entry event
    the SpMV would read a half-built vector.  Also through a declared non-default caller stream."""
    import scipy.sparse as sp
    S = km.solvers
    n = 200000
    M = (sp.diags([-1.0, -1.0], [-1, 1], shape=(n, n)) + sp.diags(np.full(n, 4.0))).tocsr()
    M.sort_indices()
    comm = _comm(km, n)
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    want = M @ np.ones(n)
    lib = km.lib.load()
    for use_side_stream in (False, True):
        side = torch.cuda.Stream() if use_side_stream else None
        if side is not None:
            km.lib.check(lib.kmcf_comm_set_caller_stream(comm.handle, side.cuda_stream), "set_caller_stream")
        with torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream()):
            p = torch.zeros(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            for _ in range(4000):
                p += 0.00025                     # 4000 launches queued, not yet executed
            Ap = torch.empty_like(p)
            mat.spmv(p, Ap)
        got = Ap.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-9)
    km.lib.check(lib.kmcf_comm_set_caller_stream(comm.handle, None), "set_caller_stream")
    mat.close()
    comm.close()
