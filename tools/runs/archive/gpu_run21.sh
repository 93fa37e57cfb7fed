#!/bin/bash
# round-1 run 21: zero-copy R1CS views (pinned staging actually used), per-MSM breakdown of a serial prove
set -x
mkdir -p gpurun_out/r21
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_TIMING=1 timeout 600 python bench.py --steps 10 > gpurun_out/r21/bench.txt 2> gpurun_out/r21/bench_err.txt
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 300 python tools/prove_bench.py 3 > gpurun_out/r21/prove_serial.txt 2> gpurun_out/r21/prove_serial_err.txt
timeout 300 python -m pytest tests/test_gpu_mpn_prove.py tests/test_gpu_groth16.py -x -q > gpurun_out/r21/pytest_prove.txt 2>&1
echo finished
