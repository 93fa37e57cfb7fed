#!/bin/bash
# round-1 run 28: software-pipelined stand-alone MSM (two window groups over two streams)
set -x
mkdir -p gpurun_out/r28
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -x -q > gpurun_out/r28/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r28/pytest.txt
timeout 600 python bench.py > gpurun_out/r28/bench.txt 2> gpurun_out/r28/bench_err.txt
timeout 300 python tools/sweep.py r28 > gpurun_out/r28/sweep.txt 2>&1
echo finished
