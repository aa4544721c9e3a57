"""GPU: a current solve whose RESULT is a property of the device (BASELINE config 3: "40nm_crossbar potential + current
+ heat").  The routine full-size test (test_gpu_fullsize.py::test_full_size_current_and_heat) keeps the vacancies-only
tunnel set, whose macroscopic current is zero +- the solver's residual; here the reference's own contact window
(get_is_tunnel_mpi, src/initialize_sparsity_T.cu:618-654, window at :645) on the synthetic crossbar of 4 x 4 cells with
a vacancy filament at one crossing (structure.synth_crossbar_40nm(filament=)): 17 722 tunnel points, 44 % dense -- the
authors' shape class (main_test_cg_split.cpp:1030-1035) -- and a current of 3.9e-3 (in units of G0 V), four orders above
what the stopping rule leaves undetermined.  No reference result exists for it (parity unpinned: the reference's golden
run never executes the current solver); what is held are relations the device's physics and Kirchhoff's law fix:

* the current flows in, and the injection-side sum the reference prints (get_imacro_sparse,
  src/current_solver_gpu.cu:501-542) equals the loop-side current loop_G (Vd - (m[1] - m[0])) to 1e-6 relative at the
  tolerance 1e-15 N, 1e-8 at 1e-18 N (they differ by the source node's residual);
* the implementations of the same operator agree on it: tunnel block as dense symmetric tiles (the default at this
  density) / jagged symmetric tiles (entries only: identical sums) / bitmap + packed values, conduction-band edge from
  the PCG form / from the reference's literally scaled form (solve_sparse_CG_Jacobi, src/iterative_solvers_gpu.cu:716-887) -- to 1e-8 relative at 1e-18 N (measured 5e-10 ...
  2e-9: different summation orders under a stopping rule that bounds the current's error by 2.4e-7 of it);
* it responds to the element state: the same device without the filament carries 0.55 % less."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q = 1.60217663e-19


def _device(km, filament, comm=None, d=None):
    import torch
    S = km.solvers
    d = d if d is not None else km.structure.synth_crossbar_40nm(tiles=4, filament=filament)
    N, NL = d["N"], d["N_contact"]
    if comm is None:
        comm = S.KMC_comm(N - 2 * NL, N + 1, N, N)
        comm.connect()
    buf = S.GPUBuffers(N, d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"], d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types, buf.num_metal_types_,
                        comm.counts_events, comm.displs_events, comm)
    el = d["element"]
    N_atom = int(((el != 0) & (el != 1)).sum())
    comm.counts_T, comm.displs_T = comm.partition(N_atom + 1, comm.size_T)
    return dict(S=S, d=d, comm=comm, buf=buf, N_atom=N_atom, torch=torch)


def _current(dev, tol, dense=None, cb_scaled=None, first=False, touch_env=True, cb=None):
    """CB edge -> T assembly (the reference's window) -> solve; returns (I_macro, loop-side current, stats, info, bound).
    touch_env False: the environment is the caller's (a rank thread of a group must not change it under the others)."""
    S, d, buf, comm, N_atom = dev["S"], dev["d"], dev["buf"], dev["comm"], dev["N_atom"]
    N, NL = d["N"], d["N_contact"]
    high_G, low_G, loop_G = 1e5 * d["high_G"], d["low_G"], 1e7 * d["high_G"]
    G0 = 2 * 3.8612e-5 * 1e-5
    env = {"KMCF_SUB_DENSE": dense, "KMCF_CB_SCALED": cb_scaled} if touch_env else {}
    saved = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if cb is not None:                             # (a rank of a group: the band edge is a one-rank solve, as in the reference)
            buf.site_CB_edge = dev["torch"].as_tensor(cb, device="cuda")
        else:
            if buf.site_CB_edge is not None:
                buf.site_CB_edge.zero_()
            st_cb = S.update_CB_edge_gpu_sparse(buf, N, NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"], d["nn_dist"], len(d["metals"]))
            assert st_cb["converged"] == 1
        if first:
            S.initialize_sparsity_T(buf, d["pbc"], d["nn_dist"], NL, NL, 10, comm)
        prm = S.current_params(d["Vd"], high_G, low_G, loop_G, G0, Q * 0.01, 0.85 * 9.11e-31, 1.6)     # default window = the reference's
        S.t_assemble(buf, prm)
        info = S.t_info(buf)
        buf.atom_virtual_potentials.zero_()
        buf.site_power.zero_()
        im, st = S.update_power_gpu_sparse_dist(buf, NL, NL, 10, d["Vd"], high_G, low_G, loop_G, G0, Q * 0.01, d["nn_dist"], 0.85 * 9.11e-31, 1.6,
                                                len(d["metals"]), False, True, 1.0, cg_tolerance=tol * N_atom, cg_max_iterations=40000)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    m = buf.atom_virtual_potentials.cpu().numpy()
    i_loop = loop_G * (d["Vd"] * G0 - (m[1] - m[0]))
    bound = G0 * np.sqrt((loop_G + NL * high_G) * st["rz"])          # |I_inj - I_loop| = G0 |r_1| <= G0 sqrt(T_11 r.z)
    return im, i_loop, st, info, bound


def test_conducting_crossbar_current_is_a_property_of_the_device(km):
    dev = _device(km, 4.0)
    try:
        el = dev["d"]["element"]
        # reference tolerance class
        im, il, st, info, bound = _current(dev, 1e-15, first=True)
        print("conducting 4 x 4: %d tunnel points (%d vacancies), %.0f %% dense, stored as %s (%.2f GB); tol 1e-15 N: %d iterations, %.1f ms, "
              "I_macro %.10e, loop side %.10e, residual bound %.1e" % (info["tunnel_points"], int((el == 2).sum()), 100.0 * info["nnz_tunnel"] / info["tunnel_points"] ** 2,
                                                                       ("bitmap", "dense symmetric tiles", "jagged symmetric tiles")[info["tunnel_dense"]], info["tunnel_bytes"] * 1e-9,
                                                                       st["iterations"], st["ms_solve"], im, il, bound))
        assert info["tunnel_points"] == 17722 and 0.40 <= info["nnz_tunnel"] / info["tunnel_points"] ** 2 <= 0.48
        assert st["converged"] == 1 and im > 3e-3 and il > 3e-3                      # the current flows in, orders above ...
        assert bound <= 1e-3 * im                                                    # ... what the stopping rule leaves open
        assert abs(im - il) <= bound * 1.01 and abs(im - il) <= 1e-6 * im, (im, il, bound)
        pw = dev["buf"].site_power.cpu().numpy()
        metal = np.isin(el, dev["d"]["metals"])
        assert np.all(pw[metal] == 0) and np.all(pw >= 0) and pw.max() > 0
        # tight solves: the same current from every implementation of the operator
        res = {}
        for name, kw in (("tiles", dict(dense="1")), ("jagged tiles", dict(dense="2")), ("bitmap", dict(dense="0")),
                         ("tiles, CB edge scaled form", dict(dense="1", cb_scaled="1"))):
            i2, l2, s2, inf2, b2 = _current(dev, 1e-18, **kw)
            assert s2["converged"] == 1 and int(inf2["tunnel_dense"]) == int(kw["dense"])
            assert abs(i2 - l2) <= max(b2 * 1.01, 1e-25) and abs(i2 - l2) <= 1e-8 * i2, (name, i2, l2, b2)
            res[name] = i2
            print("  %-28s tol 1e-18 N: %d iterations, %.1f ms (%.3f ms per iteration, %.2f GB per application), I_macro %.12e (loop side %.12e)"
                  % (name, s2["iterations"], s2["ms_solve"], s2["ms_solve"] / s2["iterations"], inf2["tunnel_bytes"] * 1e-9, i2, l2))
        assert res["jagged tiles"] == res["tiles"]                                    # (the same sums: tests/test_gpu_tpath.py)
        for name, v in res.items():
            assert abs(v - res["tiles"]) <= 1e-8 * res["tiles"], (name, v, res["tiles"])
        assert abs(res["tiles"] - im) <= 1e-5 * im                                   # (and the looser solve found the same current)
        i_fil = res["tiles"]
    finally:
        dev["buf"].freeGPUmemory()
        dev["comm"].close()
    # the same device without the filament
    dev0 = _device(km, None)
    try:
        i0, l0, s0, inf0, b0 = _current(dev0, 1e-18, first=True)
        print("  without the filament: %d tunnel points, I_macro %.12e (%.3f %% less)" % (inf0["tunnel_points"], i0, 100.0 * (i_fil - i0) / i_fil))
        assert s0["converged"] == 1 and abs(i0 - l0) <= 1e-8 * i0
        assert 2e-3 * i_fil <= i_fil - i0 <= 2e-2 * i_fil, (i_fil, i0)
    finally:
        dev0["buf"].freeGPUmemory()
        dev0["comm"].close()


def test_conducting_crossbar_over_a_rank_group(km, monkeypatch):
    """The same conducting device over a group of two ranks (in-process, peer-to-peer transport, ONE GPU: a correctness
    test with a timing note, not a scaling measurement): the storage rule gives the group the tunnel block as dense
    symmetric tiles with their strips dealt to the ranks (17 722 points, 44 % full; kmcf_subop::spread), no longer the
    row-sliced bitmap form -- and the current is the one rank's, to the 1e-8 the other implementations of the operator
    agree to.  (The band edge it is assembled from is solved on one rank, as in the reference, and handed to the group.)"""
    import threading
    import torch
    S = km.solvers
    monkeypatch.setenv("KMCF_TRANSPORT", "p2p")
    monkeypatch.setenv("KMCF_P2P_TIMEOUT_MS", "60000")
    monkeypatch.delenv("KMCF_CB_SCALED", raising=False)
    cells = int(os.environ.get("KMCF_CONDUCT_TILES", "4"))            # (8: the full 40 nm crossbar, run once for profiles/r04/extras)
    d = km.structure.synth_crossbar_40nm(tiles=cells, filament=4.0)
    N, NL = d["N"], d["N_contact"]
    N_atom = int(((d["element"] != 0) & (d["element"] != 1)).sum())
    P = 2
    # one rank first: its current (dense tiles; this file's first test: 3.861416942e-3) and the band edge for the group
    monkeypatch.setenv("KMCF_SUB_DENSE", "1")
    one = _device(km, 4.0, d=d)
    try:
        i1 = _current(one, 1e-18, first=True, touch_env=False)[0]
        cb = one["buf"].site_CB_edge.cpu().numpy().copy()
    finally:
        one["buf"].freeGPUmemory()
        one["comm"].close()
    res = {}
    for name, dense in (("tiles", None), ("bitmap", "0")):
        if dense is None:
            monkeypatch.delenv("KMCF_SUB_DENSE", raising=False)
        else:
            monkeypatch.setenv("KMCF_SUB_DENSE", dense)
        comms = S.KMC_comm.loopback_group(N - 2 * NL, N_atom + 1, N, N, P)
        out, errs = [None] * P, []

        def work(r):
            try:
                torch.cuda.set_device(0)
                dev = _device(km, 4.0, comm=comms[r], d=d)
                try:
                    out[r] = _current(dev, 1e-18, first=True, touch_env=False, cb=cb)
                finally:
                    dev["buf"].freeGPUmemory()
            except Exception as e:  # pragma: no cover
                import traceback
                errs.append("rank %d: %s\n%s" % (r, e, traceback.format_exc()))

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
        assert not errs, "\n".join(errs)
        assert all(o is not None for o in out), "a rank did not finish"
        for c in comms:
            c.close()
        for im, il, st, info, bound in out:
            assert st["converged"] == 1 and (info["tunnel_points"] == 17722 or cells != 4)
            assert info["tunnel_dense"] == (1 if name == "tiles" else 0)
            assert im == out[0][0] and abs(im - il) <= max(bound * 1.01, 1e-25) and abs(im - il) <= 1e-8 * im
        res[name] = out
        print("  group of 2, %-6s: %d tunnel points, %d iterations, %.1f ms (%.3f ms per iteration; per rank %s GB), I_macro %.12e"
              % (name, out[0][3]["tunnel_points"], out[0][2]["iterations"], out[0][2]["ms_solve"], out[0][2]["ms_solve"] / out[0][2]["iterations"],
                 " / ".join("%.2f" % (o[3]["tunnel_bytes"] * 1e-9) for o in out), out[0][0]))
    nb = (res["tiles"][0][3]["tunnel_points"] + 63) // 64
    held = [o[3]["tunnel_bytes"] // 32768 for o in res["tiles"]]
    assert sum(held) == nb * (nb + 1) // 2 and abs(held[0] - held[1]) <= 16
    for name in res:
        assert abs(res[name][0][0] - i1) <= 1e-8 * i1, (name, res[name][0][0], i1)
