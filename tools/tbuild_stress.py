#!/usr/bin/env python3
"""Stress: the in-process P-rank T pattern build (tests/test_gpu_tpath.py::test_small_device_multirank) over and over,
with device memory that was filled with junk and freed before every round -- a build that reads memory it never wrote
shows up as a halo-count mismatch (seen twice, rank 2 of 3, in full-suite runs)."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import kmcfield_amd as km
import test_gpu_tpath as TT
S = km.solvers
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
transport = sys.argv[3] if len(sys.argv) > 3 else "p2p"
if transport == "p2p":
    os.environ["KMCF_TRANSPORT"] = "p2p"; os.environ["KMCF_P2P_TIMEOUT_MS"] = "20000"
d = TT.small_device(seed=11)
metals = np.array([TT.TI, TT.N_EL], np.int32)
N = len(d["element"])
import kmcf_oracle as O
bad = 0
for rep in range(reps):
    junk = [torch.randint(0, 400, (1 << 22,), dtype=torch.int32, device="cuda") for _ in range(8)]   # 128 MB of plausible-looking ints
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    a = 2.5
    comms = S.KMC_comm.loopback_group(None, None, N, N, P) if False else None
    # the same sizes as the test: Nsub = atoms + 1
    el = d["element"]; Nsub = int(((el != 0) & (el != 1)).sum()) + 1 if False else None
    T = O.TSystem(d["xyz"], d["element"], d["charge"], d["cb"], metals, TT.PAR["nn_dist"], d["n1"], d["n1"], d["layers"], TT.PAR["Vd"], TT.PAR["high_G"],
                  TT.PAR["low_G"], TT.PAR["loop_G"], TT.PAR["tol"], TT.PAR["m_e"], TT.PAR["V0"], 0.0, 1.0) if rep == 0 else T
    comms = S.KMC_comm.loopback_group(T.Nsub, T.Nsub, N, N, P)
    errs = []
    def work(r):
        try:
            torch.cuda.set_device(0)
            buf = TT._make(km, torch, d["xyz"], d["element"], d["charge"], d["cb"], metals, d["n1"], d["layers"], comms[r])
            buf.freeGPUmemory()
        except Exception as e:
            errs.append("rank %d: %s" % (r, str(e)[:300]))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(P)]
    [t.start() for t in th]; [t.join(120) for t in th]
    if errs:
        bad += 1
        print("rep", rep, errs[0], flush=True)
    for c in comms:
        try: c.close()
        except Exception as e: print("close:", e)
print("%d of %d rounds failed (P=%d, %s)" % (bad, reps, P, transport))
