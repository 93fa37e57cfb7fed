#!/bin/bash
# round-4 run 32: the two GPU tests run 30 / 31 left out (the 2^24 production Update proof byte-equal to the oracle prover) and a short differential fuzz on the shipped build
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run32; mkdir -p $O
export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_gpu_production.py -m gpu -q -x -k "update_15_3_4" --durations=3 > $O/pytest_update.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_update.txt
tail -6 $O/pytest_update.txt
timeout 45 python tests/tools/fuzz_gpu.py 30 3131 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt
tail -3 $O/fuzz.txt | cut -c1-500
echo finished
