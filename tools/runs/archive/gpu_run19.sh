#!/bin/bash
# round-1 run 19: LDS multi-pass NTT
set -x
mkdir -p gpurun_out/r19
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py -x -q -k ntt > gpurun_out/r19/pytest_ntt.txt 2>&1; echo "rc=$?" >> gpurun_out/r19/pytest_ntt.txt
BZK_NTT_BMAX=4 timeout 300 python -m pytest tests/test_gpu_poseidon_ntt.py -x -q -k "ntt_vs_oracle and not 21 and not 22 and not 16 and not 17" > gpurun_out/r19/pytest_ntt_bmax4.txt 2>&1; echo "rc=$?" >> gpurun_out/r19/pytest_ntt_bmax4.txt
timeout 600 python tools/sweep.py r19 > gpurun_out/r19/sweep.txt 2>&1
echo finished
