#!/bin/bash
# round-1 run 35: equal-length task splitting; what a rank of the 2/4/8-GPU window-sharded MSM does, timed on one GPU
set -x
mkdir -p gpurun_out/r35
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_tree4.py tests/test_gpu_mpn_prove.py -x -q > gpurun_out/r35/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r35/pytest.txt
timeout 600 python tools/sweep.py r35 > gpurun_out/r35/sweep.txt 2>&1
echo finished
