#!/bin/bash
# the row-per-lane kernel beyond the Infinity Cache: synthetic crossbars of 10 x 10 and 12 x 12 tiles
cd $GRAFT_REPO_ROOT/accelerated-*/csrc
for NT in 0 1; do
  touch kmcf_spmv.hip
  if [ $NT = 1 ]; then make EXTRA=-DKMCF_SELL_NT_LAB > /dev/null 2>&1; else make > /dev/null 2>&1; fi
  cd $GRAFT_REPO_ROOT
  for T in 8 10 12; do echo "== NT=$NT tiles=$T"; LAB_TILES=$T timeout -k 10 400 python tools/spmv_lab.py "SELL=1" 2>&1 | grep "matrix:\|SELL=1"; done
  cd accelerated-*/csrc
done
touch kmcf_spmv.hip; make > /dev/null 2>&1
