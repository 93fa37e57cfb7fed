#!/bin/bash
set -x
mkdir -p gpurun_out/r32
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r32/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r32/pytest_gpu.txt
echo finished
