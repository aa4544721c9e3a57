#!/bin/bash
# GPU box: per-kernel durations and gaps of the CG iteration on one rank's share of the 40 nm matrix (--workload small)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PROF_DIR:-prof_small}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-classic cg1r}; do
  KMCF_CG_VARIANT=$v rocprofv3 --kernel-trace --stats -d $OUT/$v -o s --output-format csv -- \
      python3 $R/bench.py --workload ${WORKLOAD:-small} --no-cpu-baseline --no-hbm-probe --steps 300 --warmup 30 --repeats 2 > $OUT/$v.json 2> $OUT/$v.log
  f=$(ls $OUT/$v/*/*kernel_stats.csv $OUT/$v/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v"; head -12 $f | cut -d, -f1-8
done
