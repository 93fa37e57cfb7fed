#!/bin/bash
# round-1 run 62: the round's final build once more end to end: full GPU suite, smoke, default bench, kernel trace of the headline
set -x
mkdir -p gpurun_out/r62
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r62/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r62/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r62/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r62/smoke.txt
timeout 600 python bench.py > gpurun_out/r62/bench.txt 2> gpurun_out/r62/bench_err.txt; echo "bench rc=$?" >> gpurun_out/r62/smoke.txt
tail -3 gpurun_out/r62/pytest_gpu.txt; tail -3 gpurun_out/r62/smoke.txt; cut -c1-330 gpurun_out/r62/bench.txt
echo finished
