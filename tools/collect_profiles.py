#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_bench.sh run (gpurun_out/<dir>) into profiles/<round>/ and rewrite
profiles/spmv_traffic.json, which bench.py reads for roofline.traffic.

    python tools/collect_profiles.py gpurun_out/prof_r02 r02"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(src + "/stats/*kernel_stats.csv") + glob.glob(src + "/stats/*/*kernel_stats.csv")
    shutil.copy(stats[0], os.path.join(dst, "bench_kernel_stats.csv"))
    line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
    open(os.path.join(dst, "bench_line.json"), "w").write(line)
    for ext in ("csv", "json"):
        shutil.copy(os.path.join(src, "pmc_spmv." + ext), os.path.join(dst, "pmc_spmv." + ext))
    bench = json.loads(line)
    tj = json.load(open(os.path.join(src, "pmc_spmv.json")))
    for e in tj["kernels"]:
        e["workload"] = bench["config"]["workload"]
        e["source"] = ("profiles/%s/pmc_spmv.{csv,json}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                       "tools/profile_bench.sh), FETCH_SIZE x2 (gfx950; factor confirmed in the same run on cg_p / cg_xr, "
                       "whose byte counts are known), KiB->bytes; tools/pmc_summary.py" % rnd)
    json.dump(tj, open(os.path.join(ROOT, "profiles", "spmv_traffic.json"), "w"), indent=1)
    print("profiles/%s written; spmv_traffic.json: %s" % (rnd, [(e["kernel"], e["corrected_bytes_per_launch"]) for e in tj["kernels"]]))


if __name__ == "__main__":
    main()
