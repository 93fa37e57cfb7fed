#!/bin/bash
set -x
mkdir -p gpurun_out/r45
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 16 32 64; do BZK_DEDUP_K=$k timeout 600 python bench.py --steps 5 --no-cpu-baseline > gpurun_out/r45/bench_k$k.txt 2> gpurun_out/r45/bench_k${k}_err.txt; done
echo finished
