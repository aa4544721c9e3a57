#!/bin/bash
# brick edge of the internal row order vs the row-per-lane SpMV
for B in ${BRICKS:-7.7 6 9 11 12.5 15}; do echo "BRICK=$B"; KMCF_BRICK=$B KMCF_SPMV_VERBOSE=1 timeout -k 10 200 python tools/spmv_lab.py "SELL=1" 2>&1 | grep "SELL=1\|row-per-lane plan\|window:"; done
