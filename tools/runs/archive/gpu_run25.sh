#!/bin/bash
set -x
mkdir -p gpurun_out/r25
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sl in 2 3 4; do BZK_BENCH_SLOTS=$sl timeout 600 python bench.py --steps 5 --no-cpu-baseline > gpurun_out/r25/bench_slots$sl.txt 2> gpurun_out/r25/bench_slots${sl}_err.txt; done
echo finished
