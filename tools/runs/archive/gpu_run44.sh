#!/bin/bash
set -x
mkdir -p gpurun_out/r44
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/soak.py 1000 > gpurun_out/r44/soak.txt 2> gpurun_out/r44/soak_err.txt
timeout 400 python tools/fuzz_gpu.py 120 11 > gpurun_out/r44/fuzz.txt 2>&1
echo finished
