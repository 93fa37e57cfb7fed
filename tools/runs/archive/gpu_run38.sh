#!/bin/bash
set -x
mkdir -p gpurun_out/r38
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python tools/fuzz_gpu.py 90 1 > gpurun_out/r38/fuzz1.txt 2>&1
timeout 400 python tools/fuzz_gpu.py 60 2 > gpurun_out/r38/fuzz2.txt 2>&1
echo finished
