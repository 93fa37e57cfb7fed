#!/bin/bash
set -x
mkdir -p gpurun_out/r55
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_worker.py -q > gpurun_out/r55/pytest_worker.txt 2>&1; echo "rc=$?" >> gpurun_out/r55/pytest_worker.txt
tail -30 gpurun_out/r55/pytest_worker.txt
echo finished
