#!/bin/bash
# run 54: differential fuzzing of the final build (incl. folded tables and every window size), 100 s
set -x
mkdir -p gpurun_out/r54
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tests/tools/fuzz_gpu.py 100 52 > gpurun_out/r54/fuzz.txt 2> gpurun_out/r54/fuzz_err.txt; echo "rc=$?" >> gpurun_out/r54/fuzz.txt
cat gpurun_out/r54/fuzz.txt; tail -5 gpurun_out/r54/fuzz_err.txt
echo finished
