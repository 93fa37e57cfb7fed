#!/bin/bash
set -x
mkdir -p gpurun_out/r36
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py -x -q > gpurun_out/r36/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r36/pytest.txt
timeout 600 python tools/sweep.py r36 > gpurun_out/r36/sweep.txt 2>&1
echo finished
