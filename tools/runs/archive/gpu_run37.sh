#!/bin/bash
set -x
mkdir -p gpurun_out/r37
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r37/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r37/pytest_gpu.txt
timeout 300 python -m pytest tests/test_golden_cpu.py -q > gpurun_out/r37/pytest_golden_cpu_on_gpubox.txt 2>&1
timeout 600 python tools/prove_production.py 4 3 0 > gpurun_out/r37/production.txt 2>&1
echo finished
