#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_r04.sh run (gpurun_out/prof_r04) into profiles/r04/ and rewrite
profiles/spmv_traffic.json, which bench.py reads for roofline.traffic / roofline_csr.traffic / roofline_hbm.traffic
(each entry stamped with the hash of the kernel source it was measured on: tools/pmc_summary.py)."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04")
DST = os.path.join(ROOT, "profiles", "r04")


def stats_csv(d):
    f = glob.glob(d + "/*kernel_stats.csv") + glob.glob(d + "/*/*kernel_stats.csv")
    return f[0]


def main():
    os.makedirs(os.path.join(DST, "extras"), exist_ok=True)
    shutil.copy(stats_csv(SRC + "/stats"), DST + "/bench_kernel_stats.csv")
    shutil.copy(stats_csv(SRC + "/statsbig"), DST + "/hbm_probe_kernel_stats.csv")
    line = [l for l in open(SRC + "/bench_line.json") if l.startswith("{")][-1]
    open(DST + "/bench_line_under_rocprof.json", "w").write(line)
    for n in ("pmc_spmv.csv", "pmc_spmv.json", "pmc_spmv_hbm.csv", "pmc_spmv_hbm.json"):
        shutil.copy(os.path.join(SRC, n), os.path.join(DST, n))
    bench = json.loads(line)
    kernels = []
    for name in ("pmc_spmv.json", "pmc_spmv_hbm.json"):
        tj = json.load(open(os.path.join(SRC, name)))
        for e in tj["kernels"]:
            e.setdefault("workload", bench["config"]["workload"])
            e["source"] = ("profiles/r04/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_r04.sh), "
                           "FETCH_SIZE x2 (gfx950; factor confirmed in the 40 nm run on cg_p / cg_xr, whose byte counts are known), "
                           "KiB->bytes; tools/pmc_summary.py" % name.replace(".json", ".{csv,json}"))
            kernels.append(e)
    json.dump(dict(kernels=kernels, calibration=json.load(open(SRC + "/pmc_spmv.json"))["calibration"]),
              open(os.path.join(ROOT, "profiles", "spmv_traffic.json"), "w"), indent=1)
    ex = SRC + "/extras"
    for f in sorted(glob.glob(ex + "/*.json") + glob.glob(ex + "/*.txt")):
        if os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(DST, "extras", os.path.basename(f)))
    for d, n in (("tpath", "tpath_40nm_kernel_stats.csv"), ("events", "kmc_step_40nm_kernel_stats.csv"),
                 ("small_resident", "bench_small_cg1r_resident_kernel_stats.csv"), ("small_cg1r", "bench_small_cg1r_loop_kernel_stats.csv")):
        try:
            shutil.copy(stats_csv(os.path.join(ex, d)), os.path.join(DST, "extras", n))
        except IndexError:
            print("no kernel stats under", d)
    print("profiles/r04 written:", [(e["kernel"][:28], e["rows"], e["corrected_bytes_per_launch"], e.get("source_sha")) for e in kernels])


if __name__ == "__main__":
    main()
