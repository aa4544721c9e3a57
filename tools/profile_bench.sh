#!/bin/bash
# GPU box: the profiles behind bench.py's roofline block.
#   1. rocprofv3 --kernel-trace --stats of the default bench run   -> gpurun_out/prof/stats/
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (never combined with other traces)
#   3. tools/pmc_summary.py -> gpurun_out/prof/pmc_spmv.{csv,json}
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o b --output-format csv -- \
    python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/stats.log
tail -1 $OUT/bench_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o x --output-format csv -- \
      python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 2 --spmv-reps 5 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
done
python3 $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_spmv
