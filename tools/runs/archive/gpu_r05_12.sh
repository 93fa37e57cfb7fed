#!/bin/bash
# round-5 run 12: the whole GPU suite on the shipped build (run 11 passed pytest a plugin flag it already had: the suite did not start), every test under its own
# timeout; then the fuzzer
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run12; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -m gpu -q --timeout=420 --durations=10 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -18 $O/pytest_gpu.txt | cut -c1-200
timeout 200 python tests/tools/fuzz_gpu.py 40 555 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt; tail -2 $O/fuzz.txt | cut -c1-400
echo finished
