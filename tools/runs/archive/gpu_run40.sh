#!/bin/bash
set -x
mkdir -p gpurun_out/r40
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in 0 16 14; do BZK_MSM_C=$c timeout 600 python bench.py --steps 5 --no-cpu-baseline > gpurun_out/r40/bench_c$c.txt 2> gpurun_out/r40/bench_c${c}_err.txt; done
echo finished
