#!/bin/bash
# GPU box: the evidence behind DESIGN.md's round-2 statements that is not part of the bench line.
#   bash tools/profile_extras.sh   -> gpurun_out/extras/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/extras
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== hardware counters of the row-per-lane SpMV"
bash $R/tools/pmc_spmv_diag.sh "SELL=1" > $OUT/pmc_sell_diag.txt 2>&1 || true
sed -i '/^LAB=/,$d' $OUT/pmc_sell_diag.txt
echo "== kernel stats of the 40 nm current solve (T path) and heat"
rocprofv3 --kernel-trace --stats -d $OUT/tpath -o t --output-format csv -- \
    python3 -m pytest $R/tests/test_gpu_fullsize.py::test_full_size_current_and_heat -x -q -s > $OUT/tpath.log 2>&1 || true
echo "== KMC loop, 40 nm, 77 K"
python3 $R/tools/kmc_loop.py --workload 40nm --T 77 --steps 3 > $OUT/kmc_loop_40nm.txt 2>&1 || true
python3 $R/tools/kmc_loop.py --workload 5nm --steps 6 > $OUT/kmc_loop_5nm.txt 2>&1 || true
echo "== a rank's share of the 40 nm matrix (1/8) and the 5 nm device on one GPU"
python3 $R/bench.py --workload small --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > $OUT/bench_small.json
python3 $R/bench.py --workload 5nm --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > $OUT/bench_5nm.json
python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_steps20.json
echo done
