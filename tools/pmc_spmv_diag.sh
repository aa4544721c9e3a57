#!/bin/bash
# GPU box: hardware counters of the SpMV kernel (one rocprofv3 --pmc pass per counter group; never combined
# with sys/hip/hsa traces).  Output: gpurun_out/pmcdiag/<group>/...counter_collection.csv + a summary.
#   bash tools/pmc_spmv_diag.sh ["KIND=1,U=8,LPR2=4"]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
SPEC=${1:-KIND=1,U=8,LPR2=4}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmcdiag
rocprofv3 -L > $R/gpurun_out/pmcdiag/counters.txt 2>&1 || true
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  echo "== pass $i: $group"
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $group -d $R/gpurun_out/pmcdiag/g$i -o x --output-format csv -- \
      python3 $R/tools/spmv_lab.py "$SPEC" > $R/gpurun_out/pmcdiag/g$i.log 2>&1 || echo "pass $i failed"
done <<EOF
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
EOF
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmcdiag/g*/")):
    for f in glob.glob(d + "*counter_collection.csv") + glob.glob(d + "*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "spmv" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print("%-40s n=%4d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
