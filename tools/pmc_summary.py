#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs (one counter per run, as the MI355X guide prescribes) into a per-kernel
table and the corrected HBM traffic of the SpMV kernel.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01/pmc_spmv

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports exactly
half of the bytes of a coalesced streaming read -> doubled.  The factor is calibrated in this very run on
kernels with a known byte count (cg_p_kernel reads 4 vectors -- r, 1/diag, p, x -- and writes 2, cg_xr_kernel reads
3 -- Ap, r, 1/diag -- and writes 1; round 1: 3 / 5+2)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def kernel_source_sha():
    """What bench.py compares before it reports a committed traffic figure: the SpMV kernels' source as profiled."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*_amd", "csrc", "kmcf_spmv.hip")) + glob.glob(os.path.join(root, "*_amd", "csrc", "kmcf_internal.hpp"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load(d):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    n_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 1597080
    workload = sys.argv[5] if len(sys.argv) > 5 else None
    vec = n_rows * 8.0
    rows = []
    for k in sorted(fetch):
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((short, fetch[k][0], fetch[k][1], write.get(k, (0, 0.0))[1]))
    with open(out + ".csv", "w") as f:
        f.write("kernel,calls,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean\n")
        for r in rows:
            f.write("%s,%d,%.1f,%.1f\n" % r)
    def get(name):
        return [r for r in rows if r[0].startswith(name)]
    cal = {}
    if get("cg_p_kernel") and get("cg_xr_kernel"):          # (absent from runs that only launch the SpMV)
        p = get("cg_p_kernel")[0]
        xr = get("cg_xr_kernel")[0]
        cal = dict(cg_p_fetch_ratio=round(p[2] * 1024 / (4 * vec), 4), cg_xr_fetch_ratio=round(xr[2] * 1024 / (3 * vec), 4),
                   cg_xr_write_ratio=round(xr[3] * 1024 / (1 * vec), 4))
    # every SpMV kernel of the run (the bench line times the plan's kernel and, re-planned, the CSR stream kernel)
    kernels = []
    for sp in get("spmv_"):
        if sp[1] < 3:
            continue
        corrected = 2.0 * sp[2] * 1024 + sp[3] * 1024
        e = dict(kernel=sp[0], calls=sp[1], fetch_kib=sp[2], write_kib=sp[3], fetch_factor=2.0,
                 corrected_bytes_per_launch=int(corrected), rows=n_rows, source_sha=kernel_source_sha())
        if workload:
            e["workload"] = workload
        kernels.append(e)
    res = dict(kernels=kernels, calibration=cal)
    json.dump(res, open(out + ".json", "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
