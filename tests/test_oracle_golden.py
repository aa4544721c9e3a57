"""CPU: the oracle against the reference's golden 5 nm output and the committed derived vectors."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "k5nm_golden.json")) as f:
        return json.load(f)


def test_structure_counts(dev5, golden):
    el = dev5["element"]
    assert dev5["N"] == 37650 == golden["N"]            # structures/5nm_device/reordered_device_5.xyz:1
    assert int((el == 2).sum()) == 400                  # snapshot_init.xyz: 400 vacancies
    assert dev5["N_contact"] == 576                     # parameters.txt num_atoms_first_layer


def test_pattern_counts_and_brute_force(oracle, dev5, ref5, golden):
    ks = ref5["ks"]
    assert ks.n == golden["N_interface"] == 36498
    assert ks.nnz == golden["nnz"] == 940008
    assert len(ks.left_col) == golden["left_nnz"] == 2784 and len(ks.right_col) == golden["right_nnz"] == 2784
    deg = np.diff(ks.row_ptr)
    assert deg.min() == 4 and deg.max() == 53           # off-diagonal degree 3..52 + the diagonal
    assert np.bincount(deg).tolist() == golden["degree_hist"]
    assert int(ks.col.astype(np.int64).sum()) == golden["col_checksum"]
    # the cell-list pattern equals the reference's brute-force scan (iterative_solvers_gpu.cu:96-157)
    x, y, z = dev5["xyz"][:, 0], dev5["xyz"][:, 1], dev5["xyz"][:, 2]
    NL = dev5["N_contact"]
    for r0 in (0, 17000, ks.n - 300):
        rp, c = oracle.pattern(x, y, z, dev5["lattice"], 0, 3.5, 300, ks.n, NL + r0, NL, brute=True)
        assert np.array_equal(c, ks.col[ks.row_ptr[r0]:ks.row_ptr[r0 + 300]])
    rp, c = oracle.pattern(x, y, z, dev5["lattice"], 0, 3.5, 200, NL, NL + 100, 0, brute=True)
    assert np.array_equal(c, ks.left_col[ks.left_row_ptr[100]:ks.left_row_ptr[300]])
    # every row holds its diagonal, the pattern is structurally symmetric
    import scipy.sparse as sp
    M = sp.csr_matrix((np.ones(ks.nnz), ks.col, ks.row_ptr), shape=(ks.n, ks.n))
    assert (M.diagonal() == 1).all() and (M != M.T).nnz == 0


def test_pattern_pbc_cell_vs_brute(oracle):
    """pbc=1 minimum image in y,z (gpu_solvers.h:290-309): oracle's pbc path is the brute-force loop."""
    rng = np.random.default_rng(0)
    n = 400
    L = np.array([30.0, 12.0, 12.0])
    xyz = rng.random((n, 3)) * L
    rp, c = oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 1, 3.5, n, n, 0, 0)
    rp0, c0 = oracle.pattern(xyz[:, 0], xyz[:, 1], xyz[:, 2], L, 0, 3.5, n, n, 0, 0)
    assert len(c) > len(c0)       # wrap-around neighbours appear
    d = xyz[:, None, :] - xyz[None, :, :]
    d[..., 1] -= L[1] * np.round(d[..., 1] / L[1])
    d[..., 2] -= L[2] * np.round(d[..., 2] / L[2])
    want = np.sqrt((d ** 2).sum(-1)) < 3.5
    got = np.zeros((n, n), bool)
    got[np.repeat(np.arange(n), np.diff(rp)), c] = True
    assert np.array_equal(got, want)


def test_neighbor_list_and_charges(ref5, golden, dev5):
    nl, ch = ref5["neigh"], ref5["charge"]
    assert int((nl >= 0).sum(1).max()) == golden["neigh_max"] == 52      # Device.cpp:59 max_num_neighbors
    assert int(nl.astype(np.int64).sum()) == golden["neigh_checksum"]
    assert int((ch != 0).sum()) == golden["charged"] == 339               # 339 of 400 vacancies charged
    assert set(np.unique(ch)) <= {0, 2, -2}
    assert int((ch.astype(np.int64) * np.arange(dev5["N"])).sum()) == golden["charge_checksum"]


def test_assembly_invariants(ref5, golden, oracle):
    A, ks = ref5["A"], ref5["ks"]
    assert A["diag"].min() == pytest.approx(golden["diag_min"], rel=1e-15)
    assert A["diag"].max() == pytest.approx(golden["diag_max"], rel=1e-15)
    assert int((A["rhs"] != 0).sum()) == golden["rhs_nonzero"] == 1152
    # rows sum to the contact conductances: K 1 = left + right (postprocessing/test_matrices.py checks)
    y = oracle.spmv(ks.row_ptr, ks.col, A["val"], np.ones(ks.n))
    np.testing.assert_allclose(y, A["left"] + A["right"], atol=1e-12)
    assert float(np.abs(y).sum()) == pytest.approx(golden["spmv_ones_abs_sum"], rel=1e-12)
    import scipy.sparse as sp
    M = sp.csr_matrix((A["val"], ks.col, ks.row_ptr), shape=(ks.n, ks.n))
    assert abs(M - M.T).max() == 0
    # per-block diagonal summation order (P ranks) changes the diagonal by a few ulp at most
    A4 = oracle.assemble_K(ks, *ref5["asm_args"], P=4) if "asm_args" in ref5 else None
    if A4 is not None:
        np.testing.assert_allclose(A4["diag"], A["diag"], rtol=1e-15)


def test_pcg_golden(ref5, golden):
    assert ref5["iters"] == golden["pcg_iterations"] == 317
    assert ref5["relres"] == pytest.approx(golden["pcg_relres"], rel=1e-9)
    assert ref5["relres"] <= ref5["tol"]
    assert ref5["x"].min() == pytest.approx(golden["x_min"], abs=1e-9)
    assert ref5["x"].max() == pytest.approx(golden["x_max"], abs=1e-9)
    assert abs(ref5["x"]).max() <= 2.5 + 1e-7          # |V| <= Vd/2 (maximum principle, up to the CG tolerance)


def test_pcg_40_iterations_golden(oracle, ref5, golden):
    A, ks = ref5["A"], ref5["ks"]
    x40, it, rel = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], ref5["tol"], 40)
    assert it == 40
    np.testing.assert_allclose(x40[golden["x40_sample_idx"]], golden["x40_sample"], rtol=0, atol=1e-12)
    assert rel == pytest.approx(golden["pcg40_relres"], rel=1e-9)


def test_reference_snapshot_pin(oracle, dev5, ref5):
    """Loose end-to-end pin on the reference's own golden output: boundary solve + 20 A pairwise term
    vs column 5 of expected_output/Results_5.000000/snapshot_6.xyz (6 significant digits, KMC step 6)
    on the interface sites whose element did not change in the 6 KMC steps."""
    d = dev5
    NL = d["N_contact"]
    pc = oracle.poisson_gridless(d["xyz"], ref5["charge"], d["sigma"], d["k"])
    tot = pc.copy()
    tot[NL:NL + ref5["ks"].n] += ref5["x"]
    idx = np.arange(NL, NL + ref5["ks"].n)
    same = (d["element_snap6"] == d["element"])[idx]
    err = np.abs(tot[idx] - d["potential_snap6"][idx])[same]
    assert same.sum() == 36482
    assert np.median(err) <= 1e-5 and np.percentile(err, 90) <= 1e-3     # measured 2.3e-6 / 3.0e-4
    # contacts print 0 in the snapshot (boundary re-fix commented out, kmc_main.cpp:567-573)
    assert np.all(d["potential_snap6"][:NL] == 0)


def test_oracle_reproduces_reference_trajectory(oracle, km, dev5, ref5):
    """The strong pin of the oracle: the whole per-step loop of src/kmc_main.cpp:328-500 restated on the CPU
    (charges, K assembly, PCG, pairwise term, event step with std::mt19937(1)) reproduces the reference's
    expected_output: the six "KMC time is:" lines (< 2e-3 relative), the exit after six steps, and the element
    of every site of snapshot_6.xyz (the same eight events)."""
    d = dev5
    N, NL = d["N"], d["N_contact"]
    ks, nl = ref5["ks"], ref5["neigh"]
    layers = km.structure.LAYERS
    lay = km.solvers.site_layers(d["xyz"][:, 0], layers)
    el = d["element"].copy()
    ch = np.zeros(N, np.int32)
    xb = np.zeros(ks.n)
    g = oracle.mt_state(km.structure.RND_SEED_KMC)
    kmc_time, times, nev = 0.0, [], 0
    while kmc_time < d["t_switch"] and len(times) < 12:
        ch = oracle.update_charge(el, ch, nl, d["metals"])
        A = oracle.assemble_K(ks, el, ch, d["metals"], d["high_G"], d["low_G"], d["Vd"])
        xb, it, rel = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], xb, A["dinv"], 1e-14 * ks.n, 10000)
        pot = oracle.poisson_gridless(d["xyz"], ch, d["sigma"], d["k"])
        pot[NL:NL + ks.n] += xb
        t, n, log, el, ch = oracle.kmc_step(d["xyz"], nl, lay, d["T_bg"], d["freq"], d["sigma"], d["k"], pot, el, ch,
                                            layers, g, max_events=1000)
        kmc_time += t
        times.append(kmc_time)
        nev += n
    assert len(times) == 6
    np.testing.assert_allclose(times, d["kmc_times"], rtol=2e-3)
    assert nev == 8
    assert np.array_equal(el, d["element_snap6"])
    idx = np.arange(NL, N - NL)
    err = np.abs(pot[idx] - d["potential_snap6"][idx])
    assert np.median(err) <= 1e-5


def test_partition_rule(oracle):
    c, dsp = oracle.partition(36498, 8)
    assert c.tolist() == [4563, 4563, 4562, 4562, 4562, 4562, 4562, 4562] and dsp[-1] + c[-1] == 36498
    c, dsp = oracle.partition(5, 8)
    assert c.tolist() == [1, 1, 1, 1, 1, 0, 0, 0]


def test_rank_emulation_and_halo_lists(oracle, ref5, golden):
    A, ks = ref5["A"], ref5["ks"]
    for P in (2, 4, 8):
        xp, itp, relp = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"],
                                          ref5["tol"], 10000, P=P)
        assert abs(itp - ref5["iters"]) <= 3 and relp <= ref5["tol"]
        halo_cols = []
        for r in range(P):
            h = oracle.halo_lists(ks.row_ptr, ks.col, P, r)
            assert h[0]["rank"] == r
            assert [x["rank"] for x in h] == golden["halo"][str(P)][r]
            halo_cols.append(int(sum(len(x["cols"]) for x in h[1:])))
        assert halo_cols == golden["halo"][str(P) + "_halo_cols"]
    # symmetric structure: what q receives from r is what r sends to q
    P = 4
    counts, displs = oracle.partition(ks.n, P)
    H = [oracle.halo_lists(ks.row_ptr, ks.col, P, r) for r in range(P)]
    for r in range(P):
        for h in H[r][1:]:
            q = h["rank"]
            back = [b for b in H[q] if b["rank"] == r][0]
            assert np.array_equal(h["cols"] + displs[q], back["rows"] + displs[q])


def test_heat_update_closed_form(oracle):
    p = np.full(1000, 2e-9)
    T = oracle.update_temperature_global(p, 300.0, 0.5, 10.0, 3.0, 1e-6, 1e-3)
    c = 10.0 + 2e-6 / 1e-6 * 1e-3
    assert T == pytest.approx(c * (1 - 0.5 ** 3) / 0.5 + 0.5 ** 3 * 300.0, rel=1e-15)


def test_resident_order_restatement_is_the_same_recurrence(oracle, ref5):
    """oracle/kmcf_oracle_order.c: orc_pcg1_resident_order (the summation order of the register-resident solve,
    csrc/kmcf_cgr.hip) against orc_pcg1_device_order (the two-kernel loop) on a hand-made plan: the same
    single-reduction recurrence, dots added along two different trees -- iterates agree to rounding for as long as the
    recurrence is insensitive to it (40 iterations), and the converged solves sit within the summation-order spread
    of this system (conftest.py: counts 316 ... 328, potentials 5e-4 V)."""
    import numpy as np
    A, ks = ref5["A"], ref5["ks"]
    n = ks.n
    first = np.arange(0, n, 256, dtype=np.int32)
    rows = np.minimum(256, n - first).astype(np.int32)
    base = dict(rows=n, n_short=n, halo_cols=0, vec_grid=72, sell_active=1, sell_ident=1, sell_grid=(len(first) + 7) // 8 * 8, sub_grid=0,
                cg_variant=1, tile_first=first, tile_rows=rows, row_ptr=ks.row_ptr, col=ks.col, val=A["val"],
                perm=np.arange(n, dtype=np.int32))
    out = {}
    for name, extra in (("loop", dict(resident_tpb=0, resident_g1=0)), ("tpb4", dict(resident_tpb=4, resident_g1=16)),
                        ("tpb1", dict(resident_tpb=1, resident_g1=8)), ("flat", dict(resident_tpb=2, resident_g1=0))):
        plan = dict(base, **extra)
        out[name] = (oracle.pcg_device_order(plan, A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 40),
                     oracle.pcg_device_order(plan, A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 10000))
    for name in ("tpb4", "tpb1", "flat"):
        a40, a = out[name]
        b40, b = out["loop"]
        assert a40["iterations"] == b40["iterations"] == 40
        assert np.abs(a40["x"] - b40["x"]).max() <= 1e-9
        np.testing.assert_allclose(a40["rz"], b40["rz"], rtol=1e-6)
        assert a["converged"] and abs(a["iterations"] - b["iterations"]) <= 0.05 * b["iterations"]
        assert np.abs(a["x"] - ref5["x"]).max() <= 5e-4
        assert np.sqrt(a["rz"] / a["bb"]) <= ref5["tol"]
    # the reference's recurrence (two reduction points) over the resident tree against the same recurrence as the loop of kernels
    cl = dict(base, cg_variant=0)
    c40 = oracle.pcg_device_order(dict(cl, resident_tpb=2, resident_g1=0), A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 40)
    l40 = oracle.pcg_device_order(dict(cl, resident_tpb=0, resident_g1=0), A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 40)
    assert c40["iterations"] == l40["iterations"] == 40 and np.abs(c40["x"] - l40["x"]).max() <= 1e-9
    cc = oracle.pcg_device_order(dict(cl, resident_tpb=2, resident_g1=0), A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 10000)
    assert cc["converged"] and abs(cc["iterations"] - ref5["iters"]) <= 0.05 * ref5["iters"] and np.abs(cc["x"] - ref5["x"]).max() <= 5e-4
    # fixed iteration count: exactly that many, the residual after the last one reported
    cf = oracle.pcg_device_order(dict(cl, resident_tpb=2, resident_g1=0), A["rhs"], np.zeros(n), A["dinv"], ref5["tol"], 10000, fixed_iters=7)
    assert cf["iterations"] == 7 and not cf["converged"]
    # the two trees are different computations: not bit-identical (or the test would prove nothing about the tree)
    assert not np.array_equal(out["tpb4"][1]["x"], out["loop"][1]["x"]) or not np.array_equal(out["tpb1"][1]["x"], out["loop"][1]["x"])


def test_spread_tile_walk_is_the_same_operator(oracle):
    """oracle/kmcf_oracle_order.c: orc_sub_tiles_part (the dense symmetric tiles of the tunnel block with their strips
    dealt to P ranks, csrc/kmcf_tstate.hip kmcf_subop::spread): the ranks' partials add up to S x (to rounding, against
    a plain product), every tile is walked by exactly one rank (a block of ones: the partials count entries), and for
    P = 1 the walk IS the single-rank tile walk of orc_dev_spmv, bit for bit."""
    import ctypes as C
    import numpy as np
    L = oracle._order_lib()
    rng = np.random.default_rng(3)
    nt = 333                                        # 6 block rows, the last one ragged
    A = rng.standard_normal((nt, nt)) * (rng.random((nt, nt)) < 0.4)
    F = np.ascontiguousarray(np.triu(A, 1) + np.triu(A, 1).T + np.diag(rng.standard_normal(nt)))
    x = rng.standard_normal(nt)
    npad = 64 * ((nt + 63) // 64)
    for SL in (1, 2, 16):
        parts = {}
        for P in (1, 2, 3, 5):
            tot, cnt = np.zeros(npad), np.zeros(npad)
            for q in range(P):
                yp, cp = np.zeros(npad), np.zeros(npad)
                L.orc_sub_tiles_part(nt, SL, F, x, q, P, yp)
                L.orc_sub_tiles_part(nt, SL, np.ones((nt, nt)), np.ones(nt), q, P, cp)
                tot, cnt = tot + yp, cnt + cp
            np.testing.assert_array_equal(cnt[:nt], np.full(nt, float(nt)))       # every entry once, over all ranks
            assert np.all(tot[nt:] == 0) and np.abs(tot[:nt] - F @ x).max() <= 1e-12 * np.abs(F).sum(1).max() * np.abs(x).max()
            parts[P] = tot
        # P = 1 against the single-rank walk inside orc_dev_spmv (a zero neighbour matrix around it)
        rp = np.arange(nt + 1, dtype=np.int32)      # (one zero entry per row: y = 0 x + S x)
        plan = dict(rows=nt, n_short=nt, halo_cols=0, vec_grid=8, sell_active=1, sell_ident=1, sell_grid=8, sub_grid=(nt + 63) // 64, cg_variant=0,
                    tile_first=np.arange(0, nt, 256, dtype=np.int32), tile_rows=np.minimum(256, nt - np.arange(0, nt, 256)).astype(np.int32),
                    row_ptr=rp, col=np.arange(nt, dtype=np.int32), val=np.zeros(nt), perm=np.arange(nt, dtype=np.int32), boundary_grid=0, boundary_lpr=1,
                    boundary_rows=0, long_items=0)
        nzr, nzc = np.nonzero(F)
        srp = np.zeros(nt + 1, np.int32)
        np.add.at(srp, nzr + 1, 1)
        sub = dict(grid=(nt + 63) // 64, rows=np.arange(nt, dtype=np.int32), row_ptr=np.cumsum(srp).astype(np.int32), col=nzc.astype(np.int32),
                   val=F[nzr, nzc], dense=True, strip=SL)
        rk = oracle.DeviceRank(plan, sub)
        y, pap = np.zeros(nt + 1), C.c_double(0.0)
        L.orc_dev_spmv(C.byref(rk.c), np.ascontiguousarray(x), np.ascontiguousarray(x), y, 0, C.byref(pap))
        np.testing.assert_array_equal(y[:nt], parts[1][:nt])
