"""GPU: the multi-rank field solve (row partition, compact-halo SpMV with interior/boundary split,
per-rank partial dots + reduction, gathers) on ONE GPU, P ranks = P host threads of this process
joined by libkmcfield's in-process loopback transport (kmcf_comm_create_loopback).  RCCL itself
refuses two ranks on one device; this covers everything of the N>1 path except the RCCL calls,
which tests/test_gpu_forcecomm.py exercises on a 1-rank communicator.

Checked against the oracle's P-rank emulation on the reference's 5 nm device, whose shipped site
order (type-major) makes every rank a neighbour of almost every other one (hard part 3):
the worst case for the halo bookkeeping."""
import threading

import numpy as np
from conftest import TRUE_RESIDUAL_BAR
import pytest

pytestmark = pytest.mark.gpu


def _run_ranks(km, d, P, ref_charge, expect_transport="loopback", slow_rank=None):
    import torch
    S = km.solvers
    NL = d["N_contact"]
    n_if = d["N"] - 2 * NL
    comms = S.KMC_comm.loopback_group(n_if, d["N"] + 1, d["N"], d["N"], size=P, device=0)
    out = [None] * P
    errs = []

    def work(r):
        try:
            torch.cuda.set_device(0)
            comm = comms[r]
            comm.connect()
            assert comm.transport() == expect_transport, comm.transport()
            buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"],
                               d["k"], d["lattice"], d["metals"])
            S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
            S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
            S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                                buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
            charge = buf.site_charge.cpu().numpy().copy()
            mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
            info = mat.info()
            # distributed SpMV of a known global vector
            S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
            r0, nr = int(comm.displs_K[r]), int(comm.counts_K[r])
            xg = np.cos(np.arange(n_if) * 0.37) + 1.5
            p = torch.as_tensor(xg[r0:r0 + nr].copy(), device="cuda")
            Ap = torch.empty_like(p)
            mat.spmv(p, Ap)
            if slow_rank == r:
                import time
                time.sleep(1.5)              # this rank arrives late at the solve's first exchanges (see the slow-rank test)
            st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"],
                                                   d["nn_dist"], len(d["metals"]), 0)
            plan, kv = mat.sum_plan(), S.k_vectors(buf)      # what fixes this rank's summation order; the system as assembled
            S.sum_and_gather_potential(buf, NL, comm)
            res = dict(st=st, info=info, charge=charge, Ap=Ap.cpu().numpy(), r0=r0, nr=nr, plan=plan, kv=kv,
                       v=buf.site_potential_boundary.cpu().numpy().copy(),
                       tot=buf.site_potential_charge.cpu().numpy().copy())
            # second pass like a KMC step of main: pairwise rows of this rank + gather + sum, then the event step
            # with the partial totals all-gathered and the owning rank selecting (kmc_events.cu:423-470)
            S.compute_cutoff_list(comm, buf, 20.0)
            buf.site_potential_charge.zero_()
            S.poisson_gridless_gpu(buf, comm)
            S.sum_and_gather_potential(buf, NL, comm)
            res["tot2"] = buf.site_potential_charge.cpu().numpy().copy()
            lay = torch.as_tensor(S.site_layers(d["xyz"][:, 0], km.structure.LAYERS), device="cuda")
            rng = S.RandomNumberGenerator(1)
            t, nev, log = S.execute_kmc_step_mpi(comm, d["N"], comm.counts_events, comm.displs_events, 52, buf.neigh_idx,
                                                 lay, 150.0, d["freq"], d["sigma"], d["k"], buf.site_x, buf.site_y,
                                                 buf.site_z, buf.site_potential_charge, buf.site_element,
                                                 buf.site_charge, rng, km.structure.LAYERS, max_events=2000,
                                                 return_log=True)
            res.update(ev_t=t, ev_n=nev, ev_log=log, el_after=buf.site_element.cpu().numpy().copy(),
                       ch_after=buf.site_charge.cpu().numpy().copy())
            out[r] = res
            buf.freeGPUmemory()
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append("rank %d: %s\n%s" % (r, e, traceback.format_exc()))

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(180)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out), "a rank did not finish (deadlock?)"
    for c in comms:
        c.close()
    return out


@pytest.mark.parametrize("variant", ["classic", "cg1r"])
@pytest.mark.parametrize("P", [2, 4])
def test_multirank_solve_matches_oracle(km, oracle, dev5, ref5, P, variant, monkeypatch):
    # classic = the reference's recurrence (dist_conjugate_gradient.cpp:217-266); cg1r = the single-reduction
    # (Chronopoulos-Gear) recurrence multi-rank groups use by default: the same Krylov iterates in exact arithmetic.
    # Both are held to the oracle's restatement of the SAME recurrence adding in the device's order (every rank's
    # kernels, halo exchange, one partial per rank added in rank order): iteration count and solution IDENTICAL --
    # and to the oracle's natural-order P-rank emulation of the reference's recurrence through the solution bars.
    monkeypatch.setenv("KMCF_CG_VARIANT", variant)
    # the event step of a multi-rank group: replicated on every rank by default (no collective per event); the
    # reference's partitioned scheme (kmc_events.cu:423-459) stays behind KMCF_EVENTS_PARTITIONED and is exercised
    # by the "classic" runs.  Both must select the oracle's events.
    if variant == "classic":
        monkeypatch.setenv("KMCF_EVENTS_PARTITIONED", "1")
    else:
        monkeypatch.delenv("KMCF_EVENTS_PARTITIONED", raising=False)
    d = dev5
    NL = d["N_contact"]
    ks, A = ref5["ks"], ref5["A"]
    out = _run_ranks(km, d, P, ref5["charge"])
    # charges: every rank computed its rows and received the others'
    for o in out:
        assert np.array_equal(o["charge"], ref5["charge"])
    # halo bookkeeping equals the oracle's lists
    for r, o in enumerate(out):
        want = oracle.halo_lists(ks.row_ptr, ks.col, P, r)
        assert o["info"]["number_of_neighbours"] == len(want)
        assert o["info"]["halo_cols"] == sum(len(w["cols"]) for w in want[1:])
    # distributed SpMV == global SpMV
    xg = np.cos(np.arange(ks.n) * 0.37) + 1.5
    y = oracle.spmv(ks.row_ptr, ks.col, A["val"], xg)
    bound = oracle.spmv(ks.row_ptr, ks.col, np.abs(A["val"]), np.abs(xg))
    for o in out:
        sl = slice(o["r0"], o["r0"] + o["nr"])
        assert np.all(np.abs(o["Ap"] - y[sl]) <= 1e-13 * bound[sl] + 1e-300)
    # the solve: same iteration count on every rank, close to the oracle's P-rank emulation
    xo, ito, relo = oracle.pcg_jacobi(ks.row_ptr, ks.col, A["val"], A["rhs"], np.zeros(ks.n), A["dinv"], ref5["tol"],
                                      10000, P=P)
    its = {o["st"]["iterations"] for o in out}
    assert len(its) == 1
    it = its.pop()
    counts, displs = oracle.partition(ks.n, P)
    ranks = [oracle.DeviceRank(o["plan"]) for o in out]
    assert all(o["plan"]["cg_variant"] == (1 if variant == "cg1r" else 0) for o in out)
    rhs = np.concatenate([o["kv"]["rhs"] for o in out])
    dinv = np.concatenate([o["kv"]["dinv"] for o in out])
    orc = oracle.pcg_device_order_ranks(ranks, counts, displs, rhs, np.zeros(ks.n), dinv, ref5["tol"], 10000, variant=variant)
    assert it == orc["iterations"], (variant, P, it, orc["iterations"])
    assert np.array_equal(out[0]["v"][NL:-NL], orc["x"]), float(np.abs(out[0]["v"][NL:-NL] - orc["x"]).max())
    assert out[0]["st"]["rz"] == orc["rz"] and out[0]["st"]["bb"] == orc["bb"]
    assert abs(it - ito) <= 0.05 * ito, (variant, it, ito)            # natural order: another order, a nearby count
    for o in out:
        assert o["st"]["converged"] == 1 and o["st"]["relres"] <= ref5["tol"]
        # after sum_and_gather every rank holds the full interface solution
        v = o["v"]
        assert np.all(v[:NL] == 0) and np.all(v[-NL:] == 0)
        dx = np.abs(v[NL:-NL] - xo)
        assert dx.max() <= 5e-4 and np.median(dx) <= 5e-6, (dx.max(), np.median(dx))
        res = A["rhs"] - oracle.spmv(ks.row_ptr, ks.col, A["val"], v[NL:-NL])
        assert np.linalg.norm(res) / np.linalg.norm(A["rhs"]) <= TRUE_RESIDUAL_BAR
        assert np.array_equal(o["tot"], v)            # site_potential_charge (0) += boundary
    # all ranks hold bit-identical replicated solutions
    for o in out[1:]:
        assert np.array_equal(o["v"], out[0]["v"])
    # pairwise term computed row-block-wise and gathered == the oracle's, on every rank
    pw = oracle.poisson_gridless(d["xyz"], ref5["charge"], d["sigma"], d["k"], 20.0)
    for o in out:
        assert np.abs((o["tot2"] - o["v"]) - pw).max() <= 1e-12 * np.abs(pw).max()
        assert np.array_equal(o["tot2"], out[0]["tot2"])
    # event step across ranks: same events as the oracle's single-list selection fed with the same potentials
    lay = km.solvers.site_layers(d["xyz"][:, 0], km.structure.LAYERS)
    g = oracle.mt_state(1)
    t_o, n_o, log_o, el_o, ch_o = oracle.kmc_step(d["xyz"], ref5["neigh"], lay, 150.0, d["freq"], d["sigma"], d["k"],
                                                  out[0]["tot2"], d["element"], ref5["charge"], km.structure.LAYERS, g,
                                                  max_events=2000)
    for o in out:
        assert o["ev_n"] == n_o and np.array_equal(o["ev_log"], log_o)
        assert o["ev_t"] == pytest.approx(t_o, rel=1e-12)
        assert np.array_equal(o["el_after"], el_o) and np.array_equal(o["ch_after"], ch_o)


@pytest.mark.parametrize("P", [2, 4])
def test_p2p_transport_in_process_matches_loopback_bit_for_bit(km, oracle, dev5, ref5, P, monkeypatch):
    """The device-side peer-to-peer protocol (csrc/kmcf_p2p.hip: puts into the neighbours' windows, sequence flags,
    bounded waits, rank-ordered sums) driven by the P members of an in-process group, against the host-synchronous
    loopback transport: same partition, same kernels, so charges, solution, iteration count and events must be
    IDENTICAL, bit for bit (both transports add the ranks' partial dot products in rank order)."""
    monkeypatch.setenv("KMCF_CG_VARIANT", "cg1r")
    monkeypatch.setenv("KMCF_CG_RESIDENT", "0")             # (the loop of kernels on both transports; the resident launch: next test)
    monkeypatch.delenv("KMCF_EVENTS_PARTITIONED", raising=False)
    monkeypatch.delenv("KMCF_TRANSPORT", raising=False)
    base = _run_ranks(km, dev5, P, ref5["charge"])
    monkeypatch.setenv("KMCF_TRANSPORT", "p2p")
    monkeypatch.setenv("KMCF_P2P_TIMEOUT_MS", "20000")       # ranks are Python threads here: their host-side set-up can be seconds apart
    p2p = _run_ranks(km, dev5, P, ref5["charge"], expect_transport="p2p (in-process group)")
    for a, b in zip(base, p2p):
        assert np.array_equal(a["charge"], b["charge"])
        assert a["st"]["iterations"] == b["st"]["iterations"] and a["st"]["relres"] == b["st"]["relres"]
        assert np.array_equal(a["Ap"], b["Ap"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["tot2"], b["tot2"])
        assert a["ev_n"] == b["ev_n"] and np.array_equal(a["ev_log"], b["ev_log"]) and a["ev_t"] == b["ev_t"]


@pytest.mark.parametrize("P,shape", [(2, None), (4, None), (2, (4, 2)), (3, (4, 4))])
def test_resident_group_solve_matches_its_oracle_bit_for_bit(km, oracle, dev5, ref5, P, shape, monkeypatch):
    """A group's solve as ONE register-resident launch per rank (csrc/kmcf_cgr.hip with nranks > 1: the halo as
    {value, sequence} granules put straight into the receiver's window by the lane that owns the row, every rank's sums
    to a line per rank in every peer's window), P ranks = P host threads on the peer-to-peer transport: against the oracle
    adding in exactly that order (kmcf_oracle.pcg_resident_ranks) -- iteration count and every entry of the solution
    identical -- and against the reference's recurrence through the solution bars.  Charges, gathers and events as on
    the other transports."""
    monkeypatch.setenv("KMCF_CG_VARIANT", "cg1r")
    monkeypatch.setenv("KMCF_CG_RESIDENT", "1")
    if shape:                                          # (tiles per block, blocks per reduction group: the two-hop tree inside every rank)
        monkeypatch.setenv("KMCF_CGR_TPB", str(shape[0]))
        monkeypatch.setenv("KMCF_CGR_G1", str(shape[1]))
    monkeypatch.delenv("KMCF_EVENTS_PARTITIONED", raising=False)
    monkeypatch.setenv("KMCF_TRANSPORT", "p2p")
    monkeypatch.setenv("KMCF_P2P_TIMEOUT_MS", "20000")
    monkeypatch.setenv("KMCF_CGR_TIMEOUT_MS", "20000")
    out = _run_ranks(km, dev5, P, ref5["charge"], expect_transport="p2p (in-process group)")
    ks, A = ref5["ks"], ref5["A"]
    assert all(o["plan"]["cg_variant"] == 1 and o["plan"]["resident_tpb"] > 0 for o in out), [o["plan"]["resident_tpb"] for o in out]
    if shape:
        got = [(o["plan"]["resident_tpb"], o["plan"]["resident_g1"]) for o in out]
        print("resident shapes per rank:", got)
        assert all(g == shape for g in got), got
    counts, displs = oracle.partition(ks.n, P)
    ranks = [oracle.DeviceRank(o["plan"]) for o in out]
    rhs = np.concatenate([o["kv"]["rhs"] for o in out])
    dinv = np.concatenate([o["kv"]["dinv"] for o in out])
    orc = oracle.pcg_resident_ranks(ranks, counts, displs, rhs, np.zeros(ks.n), dinv, ref5["tol"], 10000)
    its = {o["st"]["iterations"] for o in out}
    assert its == {orc["iterations"]}, (its, orc["iterations"])
    NL = dev5["N_contact"]
    for o in out:
        assert np.array_equal(o["charge"], ref5["charge"])
        assert o["st"]["converged"] == 1 and o["st"]["rz"] == orc["rz"] and o["st"]["bb"] == orc["bb"]
        np.testing.assert_array_equal(o["v"][NL:NL + ks.n], orc["x"])            # replicated by sum_and_gather, identical to the oracle
    dx = np.abs(orc["x"] - ref5["x"])
    assert dx.max() <= 5e-4 and np.median(dx) <= 5e-6
    assert abs(orc["iterations"] - ref5["iters"]) <= 0.05 * ref5["iters"]
    res = A["rhs"] - oracle.spmv(ks.row_ptr, ks.col, A["val"], orc["x"])
    assert np.linalg.norm(res) / np.linalg.norm(A["rhs"]) <= TRUE_RESIDUAL_BAR


@pytest.mark.parametrize("n,P", [(10, 4), (3, 4), (2000, 3)])
def test_multirank_generic_matrix_small_and_empty_ranks(km, n, P):
    """Caller-supplied CSR over P loopback ranks: fewer rows than ranks (a rank owns nothing), a handful of
    rows per rank, and an uneven partition; distributed SpMV and Jacobi-PCG against the assembled matrix."""
    import scipy.sparse as sp
    import torch
    S = km.solvers
    M = sp.diags([np.full(n - 1, -1.0), np.full(n, 2.5) + 0.01 * np.arange(n), np.full(n - 1, -1.0)], [-1, 0, 1],
                 format="csr")
    if n > 100:
        M = (M + sp.diags([np.full(n - 37, -0.25)] * 2, [-37, 37])).tocsr()
    M.sort_indices()
    counts, displs = S.KMC_comm.partition(n, P)
    comms = S.KMC_comm.loopback_group(n, n, n, n, size=P, device=0)
    x = np.cos(np.arange(n) * 0.7) + 2.0
    b = np.sin(np.arange(n) * 0.3) + 1.5
    out = [None] * P
    errs = []

    def work(r):
        try:
            torch.cuda.set_device(0)
            comm = comms[r]
            comm.connect()
            r0, nr = int(displs[r]), int(counts[r])
            sub = M[r0:r0 + nr]
            mat = S.Distributed_matrix(comm, n, counts, displs, sub.indices, sub.indptr, sub.data)
            p = torch.as_tensor(x[r0:r0 + nr].copy(), device="cuda")
            Ap = torch.empty_like(p)
            mat.spmv(p, Ap)
            rr = torch.as_tensor(b[r0:r0 + nr].copy(), device="cuda")
            xs = torch.zeros(nr, dtype=torch.float64, device="cuda")
            dinv = torch.as_tensor(1.0 / M.diagonal()[r0:r0 + nr], device="cuda")
            st = S.conjugate_gradient_jacobi(mat, rr, xs, dinv, 1e-12, 500)
            out[r] = dict(Ap=Ap.cpu().numpy(), x=xs.cpu().numpy(), st=st, r0=r0, nr=nr)
            mat.close()
        except Exception as e:  # pragma: no cover
            import traceback
            errs.append("rank %d: %s\n%s" % (r, e, traceback.format_exc()))

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out), "a rank did not finish (deadlock?)"
    for c in comms:
        c.close()
    y = M @ x
    sol = np.concatenate([o["x"] for o in out])
    for o in out:
        np.testing.assert_allclose(o["Ap"], y[o["r0"]:o["r0"] + o["nr"]], rtol=1e-13, atol=1e-13)
        assert o["st"]["converged"] == 1 and o["st"]["iterations"] == out[0]["st"]["iterations"]
    assert np.abs(M @ sol - b).max() <= 1e-9


def test_p2p_slow_rank_cannot_have_its_halo_overwritten(km, oracle, dev5, ref5, monkeypatch):
    """A solve starts with TWO SpMVs and no all-reduce between them (A x0, then A z of the single-reduction loop): with
    three ranks that are all neighbours of each other (the 5 nm device's site order), a fast rank reaches its second
    put while a neighbour still waits for the late third rank's first one.  The landing zones are double-buffered by
    sequence parity and a put waits for the acknowledgement of the put before last (csrc/kmcf_p2p.hip), so the late
    rank changes nothing: bit-identical to the host-synchronous loopback transport."""
    monkeypatch.setenv("KMCF_CG_VARIANT", "cg1r")
    monkeypatch.delenv("KMCF_EVENTS_PARTITIONED", raising=False)
    monkeypatch.delenv("KMCF_TRANSPORT", raising=False)
    monkeypatch.setenv("KMCF_CG_RESIDENT", "0")             # (this test is about the halo protocol of the kernel loop)
    base = _run_ranks(km, dev5, 3, ref5["charge"])
    monkeypatch.setenv("KMCF_TRANSPORT", "p2p")
    monkeypatch.setenv("KMCF_P2P_TIMEOUT_MS", "20000")
    for slow in (2, 0):
        p2p = _run_ranks(km, dev5, 3, ref5["charge"], expect_transport="p2p (in-process group)", slow_rank=slow)
        for a, b in zip(base, p2p):
            assert a["st"]["iterations"] == b["st"]["iterations"] and a["st"]["rz"] == b["st"]["rz"]
            assert np.array_equal(a["v"], b["v"]) and np.array_equal(a["Ap"], b["Ap"])
