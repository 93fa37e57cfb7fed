#!/bin/bash
# round-6 run 29: the other kernels against size - Poseidon tree 4^4 .. 4^12 leaves, NTT 2^10 .. 2^26, h stage 2^14 .. 2^24: any size cliffs?
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run29; mkdir -p $O
export TMPDIR=/tmp
for lg in 8 10 12 14 16 18 20 22 24; do timeout 200 python tools/sweep.py child tree $lg | grep '^{'; done > $O/tree_sizes.txt 2>&1
for lg in 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26; do timeout 200 python tools/sweep.py child ntt $lg | grep '^{'; done > $O/ntt_sizes.txt 2>&1
for lg in 14 16 18 19 20 21 22 23 24; do timeout 200 python tools/sweep.py child h $lg | grep '^{'; done > $O/h_sizes.txt 2>&1
cat $O/tree_sizes.txt $O/ntt_sizes.txt $O/h_sizes.txt | cut -c1-300
echo finished
