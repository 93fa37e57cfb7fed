#!/bin/bash
# round-1 run 31: lazy reduction in Fp2 products (G2), host generator changes (sparse host Poseidon, in-place worker slices)
set -x
mkdir -p gpurun_out/r31
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r31/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r31/pytest_gpu.txt
timeout 300 python tools/sweep.py r31 > gpurun_out/r31/sweep.txt 2>&1
timeout 600 python bench.py > gpurun_out/r31/bench.txt 2> gpurun_out/r31/bench_err.txt
echo finished
