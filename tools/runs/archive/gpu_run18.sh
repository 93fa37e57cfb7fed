#!/bin/bash
# round-1 run 18: sparse-partial-round Poseidon
set -x
mkdir -p gpurun_out/r18
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py -x -q > gpurun_out/r18/pytest_poseidon.txt 2>&1; echo "rc=$?" >> gpurun_out/r18/pytest_poseidon.txt
timeout 300 python tools/sweep.py r18 > gpurun_out/r18/sweep.txt 2>&1
echo finished
