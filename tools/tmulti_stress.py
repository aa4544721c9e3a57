#!/usr/bin/env python3
"""Stress: tests/test_gpu_tpath.py::test_small_device_multirank, its four variants in the suite's order, cycle after cycle
in ONE process (the history a full-suite run has when the open build flake of DESIGN 11 shows)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # in-process groups: every stream its own hardware queue (tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import kmcfield_amd as km
import kmcf_oracle as O
import test_gpu_tpath as TT


class Env:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=True): os.environ.pop(k, None)


cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 10
poison = len(sys.argv) > 2 and sys.argv[2] == "poison"      # junk in device memory the library's hipMalloc may be handed afterwards
bad = 0
for c in range(cycles):
    if poison:
        n = (2 << 30) // 4 if c == 0 else (64 << 20) // 4
        junk = [torch.randint(0, 300, (n // 2,), dtype=torch.int32, device="cuda"), (torch.rand(n // 4, dtype=torch.float64, device="cuda") * 30.0)]
        torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    for transport in ("loopback", "p2p"):
        for P in (2, 3):
            storage = ("bitmap", "tiles")[c & 1]              # (the tunnel block as row slices / as tiles dealt to the ranks, alternately)
            try:
                TT.test_small_device_multirank(km, O, torch, P, transport, storage, Env())
            except AssertionError as e:
                bad += 1
                print("cycle %d, P=%d, %s, %s:\n%s" % (c, P, transport, storage, str(e)[:3000]), flush=True)
                if os.environ.get("KMCF_DEBUG_DUMP"):
                    print("stopping at the first failure (the dumps of its builds stay)"); sys.exit(0)
print("%d failures in %d cycles" % (bad, cycles))
