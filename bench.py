#!/usr/bin/env python3
"""Benchmark of the field-solve hot path: Jacobi-PCG iterations/s on the (synthetic) 40 nm
crossbar K matrix, row-partitioned over N GPUs, plus the CSR-SpMV roofline line and a CPU
baseline.  Contract: see the task description / DESIGN.md "Measurement".

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one PCG iteration (1 SpMV with fused p.Ap, the x/r/z update with fused r.z, the
p update) on the assembled K matrix with the matrix and all vectors resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def self_launch(args):
    """Parent of a multi-rank run started as ONE command.  Never initialises a GPU: a process that has done so must
    not be replaced or forked into rank processes on this pool, so the ranks are children of a process that only
    waits.  Returns the exit code to leave with."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL and the peer-to-peer windows need it here
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []

    def relay():
        for line in child.stdout:
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    try:
        rc = child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: the %d ranks did not finish within %.0f s; ending their process group\n"
                         % (args.gpus, args.launch_timeout))
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)          # the group this call created (start_new_session), nothing else
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = 124
    t.join(timeout=5)
    if rc == 0 and not any(l.lstrip().startswith("{") and '"metric"' in l for l in lines):
        sys.stderr.write("bench.py: the ranks exited 0 without printing the result line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="40nm", choices=["40nm", "5nm", "small"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="CPU baseline iterations (0 = auto, ~10-30 s)")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5, help="timed solves of --steps iterations each; the line reports their median "
                                                           "(the authors' protocol: median of 5 after warm-up, main_test_cg.cpp:209-211)")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the beyond-cache SpMV measurement (roofline_hbm)")
    ap.add_argument("--diag-solves", type=int, default=0, help="after the timed region: time N more solves piecewise (stderr)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "p2p", "p2p-only"],
                    help="multi-rank data path: auto = peer-to-peer windows if they come up, else RCCL; p2p-only = no RCCL at "
                         "all (rehearsal of N ranks on fewer GPUs, where RCCL refuses to run)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 without WORLD_SIZE: seconds the self-launched ranks may take before they are ended")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as one plain command: this process becomes the launcher.  It touches no GPU
        # (no torch import, no library load), starts the N ranks as a CHILD process group through
        # torch.distributed.run -- the protocol of the reference's own benchmark is one MPI rank per GPU,
        # dist_iterative_test/main_test_cg.cpp:94-118 -- relays rank 0's JSON line and exits with the child's code.
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    import kmcfield_amd as km
    S = km.solvers

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU path"
    if args.transport == "p2p-only" and torch.cuda.device_count() < world:
        # rehearsal: several ranks share a GPU; their chip-filling grids take a share each (kmcf_internal.hpp)
        os.environ.setdefault("KMCF_DEVICE_SHARE", str((world + torch.cuda.device_count() - 1) // max(torch.cuda.device_count(), 1)))
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible, %d ranks); --transport p2p-only rehearses more ranks than GPUs"
                         % (rank, torch.cuda.device_count(), world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        # gloo only carries the 256-byte RCCL bootstrap id, barriers and the max-over-ranks
        # timing; the data path (halos, dot products) is inside libkmcfield: RCCL, and -- KMCF_TRANSPORT=auto --
        # the peer-to-peer transport over IPC-mapped windows when its set-up and self-test succeed
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if args.transport in ("auto", "rccl", "p2p"):
            os.environ.setdefault("KMCF_TRANSPORT", args.transport)

    # ---- workload (identical on every rank; deterministic generator) ----------------------
    t_gen = time.time()
    if args.workload == "40nm":
        d = km.structure.synth_crossbar_40nm()
    elif args.workload == "small":
        # about one eighth of the 40 nm workload: what one rank holds in the 8-GPU run
        d = km.structure.synth_crossbar_40nm(tiles=3, n_lines=1)
    else:
        d = km.structure.load_device_5nm("init")
    NL = d["N_contact"]
    n_if = d["N"] - 2 * NL
    comm = S.KMC_comm(n_if, d["N"] + 1, d["N"], d["N"], rank=rank, size=world, device=local_rank)
    if world > 1 and args.transport == "p2p-only":
        comm.connect_p2p(dist)
    else:
        comm.connect(dist if world > 1 else None)
    buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"], device=local_rank)
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types,
                        buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
    S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])
    mat = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf.K_distributed))
    info = mat.info()
    vec = S.k_vectors(buf) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    t_setup = time.time() - t_gen
    n_loc = info["rows_this_rank"]
    nnz_loc = info["nnz"]
    nnz_tot = torch.tensor([nnz_loc], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(nnz_tot)
    nnz_tot = int(nnz_tot.item())

    dev = torch.device("cuda", local_rank)
    rhs_full = None

    masters = {}

    def fresh_vectors():
        # rhs / dinv of the assembled system live in the K state; the generic CG entry takes
        # caller-owned vectors like the reference (r_local_d, x_local_d, diag_inv_local_d).  Fetched once; every solve
        # gets its own copies made on the device (inputs resident in HBM before the timed region, and no 12.8 MB
        # upload from pageable memory in front of it: after one the runtime keeps reporting the caller's stream busy,
        # and the library then orders its stream behind an event on the null stream -- 40-100 us per solve)
        if not masters:
            kv = S.k_vectors(buf)
            masters["rhs"] = torch.as_tensor(kv["rhs"], device=dev)
            masters["dinv"] = torch.as_tensor(kv["dinv"], device=dev)
            torch.cuda.synchronize()
        r = masters["rhs"].clone()
        dinv = masters["dinv"]
        x = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        return r, x, dinv

    def barrier():
        comm.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- warmup + timed CG iterations -----------------------------------------------------
    tol = 1e-14 * n_if

    def timed_solve(steps, warmup):
        r, x, dinv = fresh_vectors()
        if warmup > 0:
            barrier()            # (ranks enter a solve together: its device-side waits are bounded, KMCF_P2P_TIMEOUT_MS)
            S.conjugate_gradient_jacobi(mat, r, x, dinv, tol, 10 ** 9, fixed_iters=warmup)
        r, x, dinv = fresh_vectors()
        barrier()
        t0 = time.perf_counter()
        st = S.conjugate_gradient_jacobi(mat, r, x, dinv, tol, 10 ** 9, fixed_iters=steps)
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        assert st["iterations"] == steps, st
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), st

    # Multi-rank groups with both transports up: a short trial solve on each, the faster one runs the timed region
    # (rank 0 decides).  A transport that fails its trial (a bounded wait of the peer-to-peer protocol expiring on
    # every rank) is reported and not used.
    transports = None
    P2P_MODES = {"p2p-resident": {"KMCF_CG_RESIDENT": "1"},                      # ONE register-resident launch per solve (kmcf_cgr.hip), where every rank's tiles fit
                 "p2p": {"KMCF_CG_RESIDENT": "0", "KMCF_P2P_DIRECT": "1", "KMCF_P2P_AR": "fused"},        # 3 kernels per iteration, exchanges inside them
                 "p2p-split": {"KMCF_CG_RESIDENT": "0", "KMCF_P2P_DIRECT": "1", "KMCF_P2P_AR": "split"},  # all-reduce in a 1-block kernel of its own
                 "p2p-staged": {"KMCF_CG_RESIDENT": "0", "KMCF_P2P_DIRECT": "0"}}                         # put / wait-copy kernels on a second stream

    def set_mode(name):
        for k, v in P2P_MODES.get(name, {}).items():
            os.environ[k] = v

    if world > 1 and comm.transport().startswith("p2p"):
        has_rccl = comm.transport() == "p2p (bootstrapped over rccl)"
        transports = {}
        p2p_failed = False
        for name in (["rccl"] if has_rccl else []) + list(P2P_MODES):
            if name != "rccl" and p2p_failed:
                # the peer-to-peer error word is sticky and shared by the three protocol modes: once a bounded wait has
                # expired on ANY rank every later mode would only time out again (10 s each), so none is tried
                transports[name] = {"error": "not tried: an earlier peer-to-peer trial failed"}
                continue
            err = None
            try:
                if has_rccl:
                    comm.select_transport(0 if name == "rccl" else 1)
                set_mode(name)
                t_trial, st_trial = timed_solve(min(args.steps, 20), 3)
                if name == "p2p-resident" and mat.sum_plan(with_csr=False)["resident_tpb"] == 0:
                    err = "not applicable: a rank's tiles are not all resident at once"      # (it fell back to the kernel loop: the next candidate)
            except km.lib.KmcfError as e:
                err = str(e)[:200]
            # the ranks must agree on what happened (a KmcfError is caught per rank)
            bad = torch.tensor([1 if err else 0])
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                transports[name] = {"error": err or "failed on another rank"}
                if name != "rccl" and not (err or "").startswith("not applicable"):
                    p2p_failed = True
            else:
                transports[name] = {"trial_ms_per_step": round(t_trial * 1e3 / min(args.steps, 20), 5),
                                    "trial_rz": st_trial["rz"]}
        # every candidate runs the same recurrence (the dots differ only in the order the ranks' partial sums are
        # added, and only on RCCL): a residual that disagrees beyond rounding means an exchange delivered wrong data
        ref = next((v["trial_rz"] for v in transports.values() if "trial_rz" in v), None)
        for name, v in transports.items():
            if "trial_rz" in v and not (abs(v["trial_rz"] - ref) <= 1e-6 * max(abs(ref), abs(v["trial_rz"]), 1e-300)):
                transports[name] = {"error": "residual after the trial solve differs: %r vs %r" % (v["trial_rz"], ref)}
        ok = {k: v["trial_ms_per_step"] for k, v in transports.items() if "trial_ms_per_step" in v}
        names = list(transports)
        pick = torch.tensor([names.index(min(ok, key=ok.get)) if ok else -1])
        dist.broadcast(pick, src=0)                 # rank 0 decides
        if int(pick.item()) < 0:
            raise SystemExit("bench.py: no transport passed its trial solve: %s" % json.dumps(transports))
        best = names[int(pick.item())]
        if has_rccl:
            comm.select_transport(0 if best == "rccl" else 1)
        set_mode(best)
        transports["picked"] = best
    # K steps per solve, bracketed by barriers; `repeats` such solves, the median one is reported (all are listed)
    runs = [timed_solve(args.steps, args.warmup if i == 0 else 0) for i in range(max(1, args.repeats))]
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    elapsed, st = runs[order[len(runs) // 2]]
    all_ms = [round(r[0] * 1e3 / args.steps, 5) for r in runs]
    for _ in range(args.diag_solves):           # tuning aid: where a solve's wall time goes
        r, x, dinv = fresh_vectors()
        barrier()
        ta = time.perf_counter()
        st2 = S.conjugate_gradient_jacobi(mat, r, x, dinv, tol, 10 ** 9, fixed_iters=args.steps)
        tb = time.perf_counter()
        barrier()
        tc = time.perf_counter()
        sys.stderr.write("diag solve: call %.1f us, trailing barrier %.1f us, device %.1f us\n"
                         % ((tb - ta) * 1e6, (tc - tb) * 1e6, st2["ms_solve"] * 1e3))

    # ---- roofline of the dominant kernel: CSR SpMV (+ fused p.Ap) -------------------------
    # HIP events on the library's compute stream (kmcf_spmv_bench), this rank's rows
    mat.spmv_bench(5, True)
    ms = mat.spmv_bench(args.spmv_reps, True)
    spmv_us = ms * 1e3 / args.spmv_reps
    # Algorithmic bytes of one launch = what the kernel's matrix format makes it read and write once
    # (DESIGN.md 3.1): CSR 12 B/nnz + 20 B/row; window format 10 B/nnz (f64 value + 16-bit
    # slot) + 4 B per window column + 20 B/row; dictionary-coded window format 2 B/nnz + 4 B per window column
    # + 28 B/row (row_ptr, diagonal, x, y); row-per-lane coded format below.  csr_equivalent prices the same launch at the CSR figure.
    minfo = mat.info()
    csr_bytes = 12.0 * nnz_loc + 20.0 * n_loc
    if minfo["spmv_kind"] == 2 and minfo["spmv_coded"] == 2:
        # row-per-lane coded format: 2 B per streamed entry (off-diagonals + padding to the wave's longest row) +
        # 4 B per window column + 28 B/row (lane row, diagonal, x, y) + 48 B of descriptors per tile
        kname = "spmv_sell_kernel (row-per-lane window SpMV, dictionary-coded values, fused p.Ap)"
        alg_bytes = (2.0 * minfo["spmv_stream_entries"] + 4.0 * minfo["spmv_window_cols"] + 28.0 * n_loc
                     + 48.0 * minfo["spmv_tiles"])
    elif minfo["spmv_kind"] == 2 and minfo["spmv_coded"]:
        kname = "spmv_wcode_kernel (window SpMV, dictionary-coded values, fused p.Ap)"
        alg_bytes = 2.0 * nnz_loc + 4.0 * minfo["spmv_window_cols"] + 28.0 * n_loc
    elif minfo["spmv_kind"] == 2:
        kname = "spmv_window_kernel (window SpMV, f64 values, fused p.Ap)"
        alg_bytes = 10.0 * nnz_loc + 4.0 * minfo["spmv_window_cols"] + 20.0 * n_loc
    else:
        kname = "%s (CSR SpMV, fused p.Ap)" % ("spmv_stream_kernel" if minfo["spmv_kind"] == 1 else "spmv_vec_kernel")
        alg_bytes = csr_bytes
    achieved = alg_bytes / (spmv_us * 1e-6) / 1e9
    csr_eq = csr_bytes / (spmv_us * 1e-6) / 1e9
    # what one launch streams: its matrix format + the vectors it touches (x gathered, y written, row data)
    vec_bytes = 8.0 * n_loc * 2
    working_set = alg_bytes + vec_bytes
    mall = 256 * 2 ** 20
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": kname, "us_per_launch": round(spmv_us, 2),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "working_set_bytes": int(working_set), "fits_infinity_cache": bool(working_set < mall),
                "note": "frac = format bytes / launch time / 8 TB/s.  With the 2 B/nnz format the launch's working set "
                        "fits the 256 MiB Infinity Cache and FETCH_SIZE counts its hits, so this is an effective "
                        "bandwidth against the HBM peak, not a measured HBM fraction; the CSR kernel's figure on "
                        "SURVEY 8d's bytes is in roofline_csr",
                "csr_equivalent": {"bytes_per_launch": int(csr_bytes), "achieved": round(csr_eq, 1),
                                   "frac": round(csr_eq / HBM_PEAK_GBS, 4)}}

    # ---- the CSR SpMV the roofline target is written for (12 B/nnz + 20 B/row) ----------------------
    # the same matrix re-planned onto the CSR stream kernel (f64 values + i32 columns), timed the same way, then
    # the default plan is restored
    roofline_csr = None
    roofline_f64 = None
    if world == 1 and minfo["spmv_kind"] == 2:
        saved = {k: os.environ.get(k) for k in ("KMCF_SPMV_KIND", "KMCF_SPMV_CODED")}
        try:
            # what the planner gives a matrix WITHOUT a small value dictionary (general f64 values): the row-per-lane
            # kernel on f64 values (8 B value + 2 B window slot per streamed entry, padding included, + 4 B per window
            # column + 28 B/row + 48 B per tile), or, where that stream cannot be built, the window kernel
            # (10 B/nnz + 4 B per window column + 20 B/row)
            os.environ["KMCF_SPMV_KIND"], os.environ["KMCF_SPMV_CODED"] = "2", "0"
            i_w = mat.replan()
            if i_w["spmv_kind"] == 2 and not i_w["spmv_coded"]:
                mat.spmv_bench(5, True)
                us = mat.spmv_bench(args.spmv_reps, True) * 1e3 / args.spmv_reps
                if i_w["spmv_stream_entries"] > 0:
                    k64 = "spmv_sellv_kernel (row-per-lane window SpMV, f64 values + 16-bit slots, fused p.Ap)"
                    wb = (10.0 * i_w["spmv_stream_entries"] + 4.0 * i_w["spmv_window_cols"] + 28.0 * n_loc
                          + 48.0 * i_w["spmv_tiles"])
                else:
                    k64 = "spmv_window_kernel (window SpMV, f64 values + 16-bit slots, fused p.Ap)"
                    wb = 10.0 * nnz_loc + 4.0 * i_w["spmv_window_cols"] + 20.0 * n_loc
                roofline_f64 = {"bound": "hbm", "kernel": k64,
                                "us_per_launch": round(us, 2), "algorithmic_bytes_per_launch": int(wb),
                                "achieved": round(wb / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(wb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                "csr_equivalent": {"bytes_per_launch": int(csr_bytes), "frac": round(csr_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                                "fits_infinity_cache": bool(wb + vec_bytes < mall)}
            os.environ["KMCF_SPMV_KIND"], os.environ["KMCF_SPMV_CODED"] = "1", "0"
            mat.replan()
            if mat.info()["spmv_kind"] == 1:
                mat.spmv_bench(5, True)
                us = mat.spmv_bench(args.spmv_reps, True) * 1e3 / args.spmv_reps
                a = csr_bytes / (us * 1e-6) / 1e9
                roofline_csr = {"bound": "hbm", "kernel": "spmv_stream_kernel (CSR: f64 values + i32 columns, fused p.Ap)",
                                "us_per_launch": round(us, 2), "algorithmic_bytes_per_launch": int(csr_bytes),
                                "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None,
                                "working_set_bytes": int(csr_bytes + vec_bytes),
                                "fits_infinity_cache": bool(csr_bytes + vec_bytes < mall)}
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            mat.replan()
            S.k_assemble(buf, d["Vd"], d["high_G"], d["low_G"])      # the assembly writes the value codes again

    # ---- the product kernel beyond the caches: the same row-per-lane coded SpMV on a synthetic crossbar of 12 x 12
    # cells (3.6 M rows, ~350 MB of format: larger than the 256 MiB Infinity Cache), where its bytes really come from
    # HBM.  Generated and measured here, after the timed region; this is the HBM fraction of the kernel `value` runs on.
    roofline_hbm = None
    if world == 1 and args.workload == "40nm" and not args.no_hbm_probe and minfo["spmv_coded"] == 2:
        try:
            d2 = km.structure.synth_crossbar_40nm(tiles=12)
            NL2 = d2["N_contact"]
            comm2 = S.KMC_comm(d2["N"] - 2 * NL2, d2["N"] + 1, d2["N"], d2["N"], rank=0, size=1, device=local_rank)
            comm2.connect()
            buf2 = S.GPUBuffers(d2["N"], d2["element"], d2["xyz"][:, 0], d2["xyz"][:, 1], d2["xyz"][:, 2], 52, d2["sigma"], d2["k"],
                                d2["lattice"], d2["metals"], device=local_rank)
            S.compute_neighbor_list(comm2, buf2, d2["nn_dist"], 52)
            S.initialize_sparsity_K(buf2, d2["pbc"], d2["nn_dist"], NL2, comm2)
            S.update_charge_gpu(buf2.site_element, buf2.site_charge, buf2.neigh_idx, buf2.N_, buf2.nn_, buf2.metal_types,
                                buf2.num_metal_types_, comm2.counts_events, comm2.displs_events, comm2)
            S.k_assemble(buf2, d2["Vd"], d2["high_G"], d2["low_G"])
            mat2 = S.Distributed_matrix.from_handle(km.lib.load().kmcf_kstate_matrix(buf2.K_distributed))
            i2 = mat2.info()
            if i2["spmv_coded"] == 2:
                mat2.spmv_bench(5, True)
                us2 = mat2.spmv_bench(args.spmv_reps, True) * 1e3 / args.spmv_reps
                n2 = i2["rows_this_rank"]
                b2 = 2.0 * i2["spmv_stream_entries"] + 4.0 * i2["spmv_window_cols"] + 28.0 * n2 + 48.0 * i2["spmv_tiles"]
                a2 = b2 / (us2 * 1e-6) / 1e9
                roofline_hbm = {"bound": "hbm", "kernel": kname, "workload": d2["name"], "rows": n2, "nnz": int(i2["nnz"]),
                                "us_per_launch": round(us2, 2), "algorithmic_bytes_per_launch": int(b2),
                                "achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a2 / HBM_PEAK_GBS, 4),
                                "traffic": None, "working_set_bytes": int(b2 + 16.0 * n2),
                                "fits_infinity_cache": bool(b2 + 16.0 * n2 < mall),
                                "csr_equivalent_bytes": int(12.0 * i2["nnz"] + 20.0 * n2)}
            buf2.freeGPUmemory()
            comm2.close()
        except Exception as e:                      # a diagnostic block must not cost the line
            roofline_hbm = {"error": str(e)[:200]}

    # multi-rank diagnostic: the pieces of one distributed iteration timed separately (rank 0's clock), on every
    # transport that is up
    diag = None
    if world > 1:
        reps = 200
        diag = {"transport_used": comm.transport(), "cg_variant": os.environ.get("KMCF_CG_VARIANT", "cg1r"),
                "rccl_ranks": comm.rccl_ranks(), "devices_visible": torch.cuda.device_count(),
                "ranks_per_device": int(os.environ.get("KMCF_DEVICE_SHARE", "1"))}
        if transports is not None:
            diag["transports"] = transports
        used_p2p = comm.transport().startswith("p2p")
        has_rccl = transports is not None and "rccl" in transports
        for name, use in ((("rccl", 0), ("p2p", 1)) if has_rccl else ((comm.transport(), None),)):
            if transports is not None and "error" in transports.get(name, {}):
                continue
            try:
                if use is not None:
                    comm.select_transport(use)
                d_t = {}
                for kind, key in ((0, "allreduce3_us"), (1, "halo_exchange_us"), (2, "spmv_kernels_us")):
                    mat.comm_bench(kind, 10)
                    d_t[key] = round(mat.comm_bench(kind, reps) * 1e3 / reps, 2)
                diag[name] = d_t
            except km.lib.KmcfError as e:
                diag[name] = {"error": str(e)[:200]}
        if has_rccl:
            comm.select_transport(1 if used_p2p else 0)

    # HBM traffic of that kernel from rocprofv3 PMC runs (separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    # corrections applied by tools/pmc_summary.py); measured offline on this workload, committed under
    # profiles/, and only reported when it was taken on the same matrix
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if world == 1 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        entries = tj["kernels"] if "kernels" in tj else [tj]
        # a committed figure is only reported for the kernel source it was measured on (tools/pmc_summary.py stamps
        # the hash of csrc/kmcf_spmv.hip + kmcf_internal.hpp); after any change of those files: null until re-profiled
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        sha_now = pmc_summary.kernel_source_sha()
        for blk in (roofline, roofline_csr, roofline_hbm, roofline_f64):
            if blk is None or "kernel" not in blk:
                continue
            for e in entries:
                if e.get("rows") == blk.get("rows", n_loc) and e.get("workload") == blk.get("workload", d["name"]) and \
                        e.get("kernel", "").split("<")[0] == blk["kernel"].split(" ")[0]:
                    if e.get("source_sha") == sha_now:
                        blk["traffic"] = e["corrected_bytes_per_launch"]
                        blk["traffic_source"] = e.get("source", "profiles/spmv_traffic.json")
                    else:
                        blk["traffic_stale"] = "profiles/spmv_traffic.json was measured on kernel source %s, this is %s" % (e.get("source_sha"), sha_now)

    # ---- CPU baseline: the oracle's OpenMP PCG (same op sequence) on the host cores ---------
    cpu = None
    if vec is not None:
        import kmcf_oracle as O
        rp, col = S.k_pattern(buf, 0)
        # the GPU box gives one GPU's share of the host: 16 cores (task rules); never oversubscribe
        O.set_threads(min(16, os.cpu_count() or 1))
        n_it = args.cpu_iters
        tc = time.perf_counter()
        O.pcg_jacobi_omp(rp, col, vec["val"], vec["rhs"], np.zeros(n_loc), vec["dinv"], tol, 10 ** 9, fixed_iters=3)
        tc = time.perf_counter()                   # (first call = page-in / thread start-up, not timed)
        O.pcg_jacobi_omp(rp, col, vec["val"], vec["rhs"], np.zeros(n_loc), vec["dinv"], tol, 10 ** 9, fixed_iters=20)
        t5 = (time.perf_counter() - tc) / 21.0     # 20 iterations + the initial SpMV
        if n_it <= 0:
            n_it = int(max(5, min(20000, 15.0 / max(t5, 1e-6))))  # ~15 s of CPU work
        tc = time.perf_counter()
        O.pcg_jacobi_omp(rp, col, vec["val"], vec["rhs"], np.zeros(n_loc), vec["dinv"], tol, 10 ** 9, fixed_iters=n_it)
        tc = time.perf_counter() - tc
        cores = O.omp_threads()
        # the same on ONE thread (SURVEY 8d asks for both), a few seconds
        O.set_threads(1)
        n1 = int(max(3, min(2000, 4.0 / max(t5 * cores * 0.6, 1e-6))))
        t1 = time.perf_counter()
        O.pcg_jacobi_omp(rp, col, vec["val"], vec["rhs"], np.zeros(n_loc), vec["dinv"], tol, 10 ** 9, fixed_iters=n1)
        t1 = time.perf_counter() - t1
        model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": round(n_it / tc, 2), "unit": "iterations/s", "cores": cores, "kind": "port",
               "cpu_model": model, "host_cores_visible": os.cpu_count(),
               "one_thread": {"value": round(n1 / t1, 2), "unit": "iterations/s", "sample": "%d iterations, %.1f s" % (n1, t1)},
               "sample": "%d fixed PCG iterations of the same matrix (oracle/kmcf_oracle.c orc_pcg_jacobi_omp, "
                         "OpenMP, %.1f s)" % (n_it, tc)}

    if rank == 0:
        out = {
            "metric": "cg_iterations_per_sec", "value": round(args.steps / elapsed, 2), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 5), "higher_is_better": True,
            "repeats": len(runs), "ms_per_step_all": all_ms, "ms_per_step_min": min(all_ms), "ms_per_step_max": max(all_ms),
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": d["name"], "rows": n_if, "nnz": nnz_tot, "sites": d["N"],
                       "partition": "1-D block rows x %d" % world,
                       "transport": comm.transport() + (" / " + transports["picked"] if transports else ""), "solver": "jacobi-pcg fixed %d iterations" % args.steps,
                       "halo_cols_rank0": info["halo_cols"], "neighbours_rank0": info["number_of_neighbours"],
                       "setup_s": round(t_setup, 2), "device_ms_cg": round(st["ms_solve"], 3)},
            "roofline": roofline,
            "roofline_csr": roofline_csr,
            "roofline_hbm": roofline_hbm,
            "roofline_f64": roofline_f64,
            "cpu_baseline": cpu,
        }
        if diag is not None:
            out["diag"] = diag
        if args.workload == "5nm":
            out["data"] = "reference 5nm_device (fixture)"
        print(json.dumps(out))
    buf.freeGPUmemory()
    comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
