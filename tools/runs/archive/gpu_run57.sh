#!/bin/bash
# run 57: Fr29 square in the Poseidon S-box: parity + tree / batch timings
set -x
mkdir -p gpurun_out/r57
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_tree4.py tests/test_golden_gpu.py -q -x > gpurun_out/r57/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r57/pytest.txt
tail -3 gpurun_out/r57/pytest.txt
timeout 300 python tools/sweep.py r18 > gpurun_out/r57/sweep.txt 2>&1
cat gpurun_out/r57/sweep.txt
echo finished
