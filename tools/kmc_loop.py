#!/usr/bin/env python3
"""Full per-step loop of the reference's main (src/kmc_main.cpp:328-500) on the library: charge update,
boundary (K) solve, pairwise term, sum/gather, KMC events -- with the per-module wall times the reference
prints into output<size>_<rank>.txt ("Z - calculation time - ...").

    python tools/kmc_loop.py [--workload 5nm|40nm] [--steps 6] [--T 300]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmcfield_amd as km  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="5nm")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--T", type=float, default=300.0)
    ap.add_argument("--max-events", type=int, default=100000)
    ap.add_argument("--event-stats", action="store_true", help="locality of the selected rows (supertiles of 128 rows)")
    a = ap.parse_args()
    S = km.solvers
    d = km.structure.load_device_5nm("init") if a.workload == "5nm" else km.structure.synth_crossbar_40nm()
    N, NL = d["N"], d["N_contact"]
    t0 = time.perf_counter()
    comm = S.KMC_comm(N - 2 * NL, N + 1, N, N)
    comm.connect()
    buf = S.GPUBuffers(N, d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"],
                       d["lattice"], d["metals"])
    S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
    S.compute_cutoff_list(comm, buf, 20.0)
    S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
    layers = km.structure.LAYERS
    xs = d["xyz"][:, 0]
    if a.workload != "5nm":      # the synthetic crossbar shares the 5 nm stack along x
        xs = np.clip(xs, layers[0]["start_x"], layers[-1]["end_x"])
    lay = torch.as_tensor(S.site_layers(xs, layers), device="cuda")
    rng = S.RandomNumberGenerator(km.structure.RND_SEED_KMC)
    comm.sync()
    print("init [s] %.3f  (sites %d)" % (time.perf_counter() - t0, N))
    kmc_time = 0.0
    freq = 10e13
    for step in range(a.steps):
        def timed(f):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = f()
            comm.sync()
            torch.cuda.synchronize()
            return time.perf_counter() - t, r
        tc, _ = timed(lambda: S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_,
                                                  buf.metal_types, buf.num_metal_types_, comm.counts_events,
                                                  comm.displs_events, comm))
        tb, st = timed(lambda: S.background_potential_gpu_sparse(buf, N, NL, NL, d["Vd"], d["pbc"], d["high_G"],
                                                                 d["low_G"], d["nn_dist"], len(d["metals"]), step))
        tp, _ = timed(lambda: S.poisson_gridless_gpu(buf, comm))
        tg, _ = timed(lambda: S.sum_and_gather_potential(buf, NL, comm))
        te, ev = timed(lambda: S.execute_kmc_step_mpi(comm, N, comm.counts_events, comm.displs_events, 52, buf.neigh_idx,
                                                      lay, a.T, freq, d["sigma"], d["k"], buf.site_x, buf.site_y,
                                                      buf.site_z, buf.site_potential_charge, buf.site_element,
                                                      buf.site_charge, rng, layers, max_events=a.max_events,
                                                      return_log=True))
        kmc_time += ev[0]
        if a.event_stats and ev[1] > 0:
            # where the selection walk lands: how often a recently used supertile (128 consecutive rows of the event list)
            # is selected again -- what a small cache of row sums in LDS would hit
            import collections
            st_ = ev[2][:, 0] // 128
            line = "  events: %d, distinct supertiles %d" % (len(st_), len(set(st_.tolist())))
            for cap in (16, 64, 256):
                lru, hits = collections.OrderedDict(), 0
                for q in st_.tolist():
                    if q in lru:
                        hits += 1
                        lru.move_to_end(q)
                    else:
                        lru[q] = 1
                        if len(lru) > cap:
                            lru.popitem(last=False)
                line += ", LRU-%d hit rate %.2f" % (cap, hits / len(st_))
            print(line, flush=True)
        print("step %d: charge %.6f | boundary %.6f (%d it) | pairwise %.6f | gather %.6f | events %.6f (%d ev) | "
              "superstep %.6f | KMC time %.5e" % (step + 1, tc, tb, st["iterations"], tp, tg, te, ev[1],
                                                   tc + tb + tp + tg + te, kmc_time), flush=True)
    buf.freeGPUmemory()
    comm.close()


if __name__ == "__main__":
    main()
