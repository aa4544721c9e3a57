#!/usr/bin/env python3
"""Stress: in-process p2p groups set up and run repeatedly (intermittent set-up time-outs)."""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import kmcfield_amd as km
S = km.solvers
d = km.structure.load_device_5nm("init")
NL = d["N_contact"]; n_if = d["N"] - 2 * NL
os.environ["KMCF_TRANSPORT"] = "p2p"; os.environ["KMCF_CG_VARIANT"] = "cg1r"
os.environ["KMCF_P2P_TIMEOUT_MS"] = os.environ.get("KMCF_P2P_TIMEOUT_MS", "4000")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    comms = S.KMC_comm.loopback_group(n_if, d["N"] + 1, d["N"], d["N"], size=P, device=0)
    errs, its = [], [None] * P
    def work(r):
        try:
            torch.cuda.set_device(0)
            comm = comms[r]; comm.connect()
            buf = S.GPUBuffers(d["N"], d["element"], d["xyz"][:, 0], d["xyz"][:, 1], d["xyz"][:, 2], 52, d["sigma"], d["k"], d["lattice"], d["metals"])
            S.compute_neighbor_list(comm, buf, d["nn_dist"], 52)
            t0 = time.time()
            S.initialize_sparsity_K(buf, d["pbc"], d["nn_dist"], NL, comm)
            t1 = time.time()
            S.update_charge_gpu(buf.site_element, buf.site_charge, buf.neigh_idx, buf.N_, buf.nn_, buf.metal_types, buf.num_metal_types_, comm.counts_events, comm.displs_events, comm)
            st = S.background_potential_gpu_sparse(buf, d["N"], NL, NL, d["Vd"], d["pbc"], d["high_G"], d["low_G"], d["nn_dist"], len(d["metals"]), 0)
            its[r] = (st["iterations"], round(t1 - t0, 2))
            buf.freeGPUmemory()
        except Exception as e:
            errs.append("rank %d: %s" % (r, str(e)[:200]))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
    [t.start() for t in th]; [t.join(120) for t in th]
    print("rep", rep, "its", its, "errs", errs, flush=True)
    for c in comms:
        try: c.close()
        except Exception as e: print("close:", e)
