#!/bin/bash
set -x
mkdir -p gpurun_out/r30
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python bench.py > gpurun_out/r30/bench.txt 2> gpurun_out/r30/bench_err.txt ) 2> gpurun_out/r30/bench_time.txt
timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -k "2p22 or 2p24" > gpurun_out/r30/pytest_big.txt 2>&1
echo finished
