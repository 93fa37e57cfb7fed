#!/bin/bash
set -x
mkdir -p gpurun_out/r24
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --no-cpu-baseline > gpurun_out/r24/bench_dedup.txt 2> gpurun_out/r24/bench_dedup_err.txt
BZK_PROVE_NODEDUP=1 timeout 600 python bench.py --steps 5 --no-cpu-baseline > gpurun_out/r24/bench_nodedup.txt 2> gpurun_out/r24/bench_nodedup_err.txt
echo finished
