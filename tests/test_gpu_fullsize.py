"""GPU: the path at BASELINE.json's full size (the synthetic 40 nm crossbar bench.py runs: 1.6 M rows,
41.8 M nnz), checked through properties that do not need a reference result of that size:

* K 1 = left + right: every row of the interface Laplacian sums to its contact conductances (the diagonal is
  assembled as the sum of the row's conductances, potential_solver_gpu.cu:774-830);
* symmetry x.(K y) = y.(K x) and linearity of the SpMV;
* all SpMV kernels (coded window / stream / vec) agree;
* assembly is idempotent (same values, bit for bit) and the values are -high_G / -low_G off the diagonal;
* the Jacobi-PCG solution satisfies its own stopping rule on the TRUE residual and the discrete maximum
  principle (the potential stays between the two contact potentials);
* the event step: legal events only, vacancy / ion bookkeeping consistent with the event types, the
  generator advanced by exactly two draws per event.
"""
import os
import ctypes as C

import numpy as np
import pytest

CODED_ON = int(os.environ.get("KMCF_SPMV_CODED", "1") != "0")   # the suite is green under KMCF_SPMV_CODED=0 too

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(km):
    import torch
    assert torch.cuda.is_available()
    S = km.solvers
    d = km.structure.synth_crossbar_40nm()
    N, NL = d["N"], d["N_contact"]
    comm = S.KMC_comm(N - 2 * NL, N + 1, N, N)
    comm.connect()
    buf = S.GPUBuffers(N, d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                        buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
    S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
    mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
    yield dict(S=S, d=d, comm=comm, buf=buf, mat=mat, n=N - 2 * NL, torch=torch)
    buf.freeGPUmemory()
    comm.close()


def _spmv(full, x):
    t = full["torch"]
    p = t.as_tensor(np.ascontiguousarray(x), device="cuda")
    Ap = t.empty_like(p)
    full["mat"].spmv(p, Ap)
    return Ap.cpu().numpy()


def test_full_size_plan_and_row_sums(full, km):
    S, d, buf, mat, n = full["S"], full["d"], full["buf"], full["mat"], full["n"]
    info = mat.info()
    assert info["rows_this_rank"] == n == 1597080 and info["nnz"] == 41834706
    assert info["spmv_kind"] == 2 and info["spmv_coded"] == CODED_ON
    v = S.k_vectors(buf)
    rp, col = S.k_pattern(buf, 0)
    # off-diagonals are one of the two conductances, the diagonal is positive
    rows = np.repeat(np.arange(n), np.diff(rp))
    off = v["val"][col != rows]
    assert set(np.unique(off)) <= {-d["high_G"], -d["low_G"]}
    assert np.all(v["diag"] > 0) and np.allclose(v["dinv"] * v["diag"], 1.0, rtol=1e-15, atol=0)
    # K 1 = left + right (interface part of the row cancels against the diagonal)
    y = _spmv(full, np.ones(n))
    assert np.abs(y - (v["left"] + v["right"])).max() <= 1e-12 * v["diag"].max()
    # assembling again changes nothing
    S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
    assert np.array_equal(mat.get_values(), v["val"])


def test_full_size_spmv_symmetry_linearity_and_kernels(full, km, monkeypatch):
    n, mat = full["n"], full["mat"]
    rng = np.random.default_rng(40)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Kx, Ky = _spmv(full, x), _spmv(full, y)
    scale = np.linalg.norm(Kx) * np.linalg.norm(y)
    assert abs(np.dot(y, Kx) - np.dot(x, Ky)) <= 1e-12 * scale                      # K = K^T
    a, b = 0.75, -1.5                                                                # exact in binary
    lin = _spmv(full, a * x + b * y)
    assert np.abs(lin - (a * Kx + b * Ky)).max() <= 1e-13 * np.abs(Kx).max()
    lib = km.lib.load()
    lib.kmcf_spmv_replan.argtypes = [C.c_void_p]
    try:
        for kind in ("1", "0"):
            monkeypatch.setenv("KMCF_SPMV_KIND", kind)
            km.lib.check(lib.kmcf_spmv_replan(mat.handle), "replan")
            assert mat.info()["spmv_kind"] == int(kind)
            assert np.abs(_spmv(full, x) - Kx).max() <= 1e-13 * np.abs(Kx).max()
    finally:
        monkeypatch.delenv("KMCF_SPMV_KIND", raising=False)
        km.lib.check(lib.kmcf_spmv_replan(mat.handle), "replan")
        assert mat.info()["spmv_kind"] == 2 and mat.info()["spmv_coded"] == CODED_ON


def test_full_size_solve_properties(full):
    S, d, buf, n, t = full["S"], full["d"], full["buf"], full["n"], full["torch"]
    NL = d["N_contact"]
    buf.site_potential_boundary.zero_()
    st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"],
                                           d["nn_dist"], len(d["metals"]), 0)
    tol = 1e-14 * n
    assert st["converged"] == 1 and st["relres"] <= tol and 50 < st["iterations"] < 5000
    v = buf.site_potential_boundary.cpu().numpy()[NL:-NL]
    vec = S.k_vectors(buf)
    # the stopping rule holds for the true residual too (z = r / diag): sqrt(r.z / b.b)
    r = vec["rhs"] - _spmv(full, v)
    true_rel = np.sqrt(np.dot(r, r * vec["dinv"]) / np.dot(vec["rhs"], vec["rhs"]))
    assert true_rel <= 50 * tol, (true_rel, tol)         # recurrence vs true residual after a few hundred iterations
    # discrete maximum principle: between the contact potentials -Vd/2 and +Vd/2
    assert v.min() >= -d["Vd"] / 2 - 1e-5 and v.max() <= d["Vd"] / 2 + 1e-5
    assert v.min() < -0.4 * d["Vd"] and v.max() > 0.4 * d["Vd"]                    # and it does span them
    # solving again from the solution stops at once
    st2 = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"],
                                            d["nn_dist"], len(d["metals"]), 1)
    assert st2["iterations"] <= 2


def test_full_size_event_step_bookkeeping(full, km):
    S, d, buf, comm, t = full["S"], full["d"], full["buf"], full["comm"], full["torch"]
    NL = d["N_contact"]
    layers = km.structure.LAYERS
    xs = np.clip(d["xyz"][:, 0], layers[0]["start_x"], layers[-1]["end_x"])
    lay = t.as_tensor(S.site_layers(xs, layers), device="cuda")
    S.sum_and_gather_potential(buf, NL, comm)
    el0 = buf.site_element.cpu().numpy().copy()
    rng = S.RandomNumberGenerator(7)
    probe = S.RandomNumberGenerator(7)
    tev, nev, log = S.execute_kmc_step_mpi(comm, d["N"], comm.counts_events, comm.displs_events, 52, buf.neigh_idx, lay,
                                           77.0, 10e13, d["sigma"], d["k"], buf.site_x, buf.site_y, buf.site_z,
                                           buf.site_potential_charge, buf.site_element, buf.site_charge, rng, layers,
                                           max_events=300, return_log=True)
    assert 1 <= nev <= 300 and tev > 0
    assert set(np.unique(log[:, 2])) <= {0, 1, 2, 3}
    assert log[:, 0].min() >= 0 and log[:, 0].max() < d["N"] and log[:, 1].min() >= 0 and log[:, 1].max() < d["N"]
    el1 = buf.site_element.cpu().numpy()
    n_gen, n_rec = int((log[:, 2] == 0).sum()), int((log[:, 2] == 1).sum())
    assert (el1 == 2).sum() - (el0 == 2).sum() == n_gen - n_rec                      # vacancies
    assert (el1 == 1).sum() - (el0 == 1).sum() == n_gen - n_rec                      # oxygen ions
    # an executed pair is never selected again within the step (its events were zeroed)
    pairs = {(int(a), int(b)) for a, b in log[:, :2]}
    assert len(pairs) == nev
    for _ in range(2 * nev):
        probe.getRandomNumber()
    assert rng.getRandomNumber() == probe.getRandomNumber()                          # two draws per event
