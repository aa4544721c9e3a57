"""GPU: conduction-band-edge Laplace solve (update_CB_edge_gpu_sparse, src/potential_solver_gpu.cu:575-772)
and the single-GPU symmetric-scaled CG it uses (solve_sparse_CG_Jacobi, src/iterative_solvers_gpu.cu:716-887),
against the oracle restatement on the reference's 5 nm device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sys5cb(km, dev5):
    import torch
    assert torch.cuda.is_available()
    S = km.solvers
    d = dev5
    NL = d["N_contact"]
    comm = S.KMC_comm(d["N"] - 2 * NL, d["N"] + 1, d["N"], d["N"], rank=0, size=1, device=0)
    comm.connect()
    buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    yield dict(comm=comm, buf=buf, d=d)
    buf.freeGPUmemory()
    comm.close()


@pytest.mark.parametrize("form", ["pcg", "scaled"])
def test_cb_edge_matches_oracle(km, oracle, sys5cb, ref5, form, monkeypatch):
    """form "pcg" (default): the solve in its Jacobi-PCG form on the coded matrix (the same iteration, kmcf_cg.hip);
    "scaled": the literal form, A scaled in place.  Both against the oracle's restatement of the scaled form."""
    if form == "scaled":
        monkeypatch.setenv("KMCF_CB_SCALED", "1")
    S = km.solvers
    buf, d = sys5cb["buf"], sys5cb["d"]
    NL = d["N_contact"]
    ks = ref5["ks"]
    want, it_o, A = oracle.update_CB_edge(ks, d["element"], d["metals"], d["high_G"], d["low_G"], d["Vd"])
    st = S.update_CB_edge_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"], d["nn_dist"],
                                     len(d["metals"]))
    got = buf.site_CB_edge.cpu().numpy()
    eV = 1.60217663e-19
    assert np.all(got[:NL] == d["Vd"] / 2 * eV) and np.all(got[-NL:] == -d["Vd"] / 2 * eV)    # :746-752
    assert abs(st["iterations"] - it_o) <= max(3, 0.05 * it_o), (st["iterations"], it_o)
    # absolute stop ||r||^2 <= 1e-28 on the scaled system: both solutions are converged to ~1e-13 V
    assert np.abs(got - want).max() / eV <= 1e-9
    assert np.abs(got).max() <= d["Vd"] / 2 * eV * (1 + 1e-9)
    # the assembled CB system (the scaled form leaves the values scaled in place, like the reference's solver)
    vec = S.k_vectors(buf)
    np.testing.assert_allclose(vec["rhs"], A["rhs"], rtol=1e-14)
    if form == "scaled":
        np.testing.assert_allclose(vec["val"], A["val_scaled"], rtol=1e-12, atol=1e-300)
    else:
        np.testing.assert_allclose(vec["val"], A["val"], rtol=1e-14, atol=1e-300)


def test_solve_sparse_CG_Jacobi_generic_csr(km, oracle, torch_cuda_mod):
    """The library-level entry on a caller-supplied CSR matrix (Distributed_matrix ctor 1): SPD 2-D
    Laplacian + diagonal shift; A and rhs are scaled in place like the reference."""
    torch = torch_cuda_mod
    import scipy.sparse as sp
    S = km.solvers
    nx = 60
    n = nx * nx
    T = sp.diags([-1, 2.3, -1], [-1, 0, 1], shape=(nx, nx))
    M = (sp.kron(sp.eye(nx), T) + sp.kron(T, sp.eye(nx))).tocsr()
    M.sort_indices()
    rng = np.random.default_rng(11)
    b = rng.standard_normal(n)
    comm = S.KMC_comm(n, n, n, n)
    comm.connect()
    mat = S.Distributed_matrix(comm, n, [n], [0], M.indices, M.indptr, M.data)
    rhs = torch.as_tensor(b.copy(), device="cuda")
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    st = S.solve_sparse_CG_Jacobi(mat, rhs, x, 1e-14, 5000)
    val = M.data.copy()
    bb = b.copy()
    y = np.zeros(n)
    L = oracle.lib()
    import ctypes as C
    L.orc_solve_sparse_CG_Jacobi.restype = C.c_int
    it_o = L.orc_solve_sparse_CG_Jacobi(n, M.indptr.astype(np.int32), M.indices.astype(np.int32), val, bb, y, 1e-14, 5000)
    assert abs(st["iterations"] - it_o) <= 2
    assert np.abs(x.cpu().numpy() - y).max() <= 1e-12
    np.testing.assert_allclose(rhs.cpu().numpy(), bb, rtol=1e-14)              # rhs scaled in place (:740)
    np.testing.assert_allclose(mat.get_values(), val, rtol=1e-14)               # A scaled in place (:745)
    assert np.abs(M @ x.cpu().numpy() - b).max() <= 1e-10
    mat.close()
    comm.close()


@pytest.fixture(scope="module")
def torch_cuda_mod():
    import torch
    assert torch.cuda.is_available()
    return torch
