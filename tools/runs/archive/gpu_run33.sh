#!/bin/bash
set -x
mkdir -p gpurun_out/r33
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tree4.py -x -q > gpurun_out/r33/pytest_tree4.txt 2>&1; echo "rc=$?" >> gpurun_out/r33/pytest_tree4.txt
timeout 600 python tools/tree_bench.py > gpurun_out/r33/tree_bench.txt 2>&1
echo finished
