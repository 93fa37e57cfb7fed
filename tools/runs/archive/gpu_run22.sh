#!/bin/bash
# round-1 run 22: scalar de-duplication for the witness MSMs
set -x
mkdir -p gpurun_out/r22
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_mpn_prove.py tests/test_gpu_groth16.py -x -q > gpurun_out/r22/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r22/pytest.txt
BZK_TIMING=1 timeout 600 python bench.py --steps 10 > gpurun_out/r22/bench.txt 2> gpurun_out/r22/bench_err.txt
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 300 python tools/prove_bench.py 3 > gpurun_out/r22/prove_serial.txt 2> gpurun_out/r22/prove_serial_err.txt
BZK_PROVE_NODEDUP=1 timeout 300 python tools/prove_bench.py 3 > gpurun_out/r22/prove_nodedup.txt 2>&1
echo finished
