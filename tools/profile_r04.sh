#!/bin/bash
# GPU box: everything under profiles/r04/ (run as ONE gpurun call, then tools/collect_r04.py here).
#   bash tools/profile_r04.sh   -> gpurun_out/prof_r04/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PART=${1:-all}          # a | b | all   (two gpurun calls keep each under the call's time limit)
if [ "$PART" != "b" ]; then
echo "== bench: kernel trace + stats (the driver's command line)"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o b --output-format csv -- \
    python3 $R/bench.py --steps 20 --warmup 5 --no-hbm-probe > $OUT/bench_line.json 2> $OUT/stats.log   # (the probe launches the SAME kernel on a larger matrix: profiled separately below, so that this summary's average is the 40 nm launch)
tail -c 300 $OUT/bench_line.json; echo
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c: 40 nm matrix (coded row-per-lane kernel, f64 row-per-lane kernel, CSR stream kernel)"
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o x --output-format csv -- \
      python3 $R/bench.py --no-cpu-baseline --no-hbm-probe --steps 30 --warmup 2 --repeats 1 --spmv-reps 5 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
  echo "== pmc $c: 12 x 12 device (3.6 M rows, beyond the Infinity Cache)"
  LAB_TILES=12 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmcbig_$c -o x --output-format csv -- \
      python3 $R/tools/spmv_lab.py "SELL=1" > $OUT/pmcbig_$c.txt 2> $OUT/pmcbig_$c.log
done
python3 $R/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_spmv 1597080
python3 $R/tools/pmc_summary.py $OUT/pmcbig_FETCH_SIZE $OUT/pmcbig_WRITE_SIZE $OUT/pmc_spmv_hbm 3596760 "synthetic_40nm_crossbar(tiles=12,fill=0.52,lines=2,seed=40,bwmin)"
echo "== big device: kernel stats"
LAB_TILES=12 rocprofv3 --kernel-trace --stats -d $OUT/statsbig -o b --output-format csv -- python3 $R/tools/spmv_lab.py "SELL=1" > $OUT/statsbig.txt 2>&1
fi
if [ "$PART" = "a" ]; then echo done-a; exit 0; fi
echo "== extras"
E=$OUT/extras; mkdir -p $E
python3 $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $E/bench_steps20.json
python3 $R/bench.py --no-cpu-baseline --no-hbm-probe --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_steps300.json
# a rank's eighth of the 40 nm matrix on one GPU: reference recurrence (3 kernels), single-reduction loop (2 kernels), resident launch
KMCF_CG_RESIDENT=0 python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_small_classic.json
python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_small_classic_resident.json
KMCF_CG_VARIANT=cg1r KMCF_CG_RESIDENT=0 python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_small_cg1r_loop.json
KMCF_CG_VARIANT=cg1r python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_small_cg1r_resident.json
KMCF_CG_VARIANT=cg1r rocprofv3 --kernel-trace --stats -d $E/small_resident -o s --output-format csv -- \
    python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 --repeats 2 --spmv-reps 5 > /dev/null 2>&1
KMCF_CG_VARIANT=cg1r KMCF_CG_RESIDENT=0 rocprofv3 --kernel-trace --stats -d $E/small_cg1r -o s --output-format csv -- \
    python3 $R/bench.py --workload small --no-cpu-baseline --steps 300 --warmup 30 --repeats 2 --spmv-reps 5 > /dev/null 2>&1
# the reference's 5 nm device
KMCF_CG_RESIDENT=0 python3 $R/bench.py --workload 5nm --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_5nm_classic_loop.json
python3 $R/bench.py --workload 5nm --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_5nm_classic_resident.json
KMCF_CG_VARIANT=cg1r KMCF_CG_RESIDENT=0 python3 $R/bench.py --workload 5nm --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_5nm_cg1r_loop.json
KMCF_CG_VARIANT=cg1r python3 $R/bench.py --workload 5nm --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 > $E/bench_5nm_cg1r_resident.json
# REHEARSAL of the N > 1 command shape on the one GPU: `python bench.py --gpus 2` launches its ranks itself
cd $R && timeout -k 10 300 python3 bench.py --gpus 2 --steps 300 --warmup 30 --transport p2p-only --workload small --launch-timeout 280 2>/dev/null | tail -1 > $E/bench_2ranks_one_gpu_small.json
cd /tmp
# current path
KMCF_T_REPEAT=2 rocprofv3 --kernel-trace --stats -d $E/tpath -o t --output-format csv -- \
    python3 -m pytest $R/tests/test_gpu_fullsize.py::test_full_size_current_and_heat -x -q -s > $E/tpath.log 2>&1
grep "T 40 nm\|CB edge" $E/tpath.log > $E/tpath_40nm.txt
python3 -m pytest $R/tests/test_gpu_conducting.py -x -q -s 2>&1 | grep "conducting\|tol 1e-18\|filament\|group of\|passed\|failed" > $E/conducting_4x4.txt
KMCF_T_FULL_WINDOW=1 python3 -m pytest $R/tests/test_gpu_fullsize.py::test_full_size_current_and_heat -x -q -s 2>&1 | grep "T 40 nm\|passed\|failed" > $E/tpath_40nm_reference_window.txt
KMCF_SUB_DENSE=2 KMCF_T_FULL_WINDOW=1 python3 -m pytest $R/tests/test_gpu_fullsize.py::test_full_size_current_and_heat -x -q -s 2>&1 | grep "reference window\|passed\|failed" > $E/tpath_40nm_reference_window_jagged.txt
# KMC loop
python3 $R/tools/kmc_loop.py --workload 40nm --T 77 --steps 3 > $E/kmc_loop_40nm.txt 2>&1
python3 $R/tools/kmc_loop.py --workload 5nm --steps 6 > $E/kmc_loop_5nm.txt 2>&1
rocprofv3 --kernel-trace --stats -d $E/events -o e --output-format csv -- python3 $R/tools/kmc_loop.py --workload 40nm --T 77 --steps 1 > $E/events.log 2>&1
echo done
