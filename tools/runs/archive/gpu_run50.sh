#!/bin/bash
# run 50: window-size coverage tests (every c, witness knob in a fresh process)
set -x
mkdir -p gpurun_out/r50
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py -q -x > gpurun_out/r50/pytest_msm.txt 2>&1; echo "rc=$?" >> gpurun_out/r50/pytest_msm.txt
tail -25 gpurun_out/r50/pytest_msm.txt
echo finished
