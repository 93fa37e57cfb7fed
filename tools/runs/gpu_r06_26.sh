#!/bin/bash
# round-6 run 26: stand-alone G2 MSM at 2^17 .. 2^20 points: does the degenerate top window (c < 16) cost G2's one-level pair fold as it cost G1's?
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run26; mkdir -p $O
export TMPDIR=/tmp
for lg in 17 18 19 20; do SWEEP_REPS=4 timeout 200 python tools/sweep.py child g2res $lg; done > $O/g2_sizes.txt 2>&1
cut -c1-900 $O/g2_sizes.txt
echo finished
