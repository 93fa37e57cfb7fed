#!/bin/bash
# round-1 run 58: the round's final build once more end to end: full GPU suite, smoke, default bench, kernel trace of the headline
set -x
mkdir -p gpurun_out/r58
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r58/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r58/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r58/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r58/smoke.txt
timeout 600 python bench.py > gpurun_out/r58/bench.txt 2> gpurun_out/r58/bench_err.txt; echo "bench rc=$?" >> gpurun_out/r58/smoke.txt
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r58/trace -- $CMD > gpurun_out/r58/trace.log 2>&1
T=$(find gpurun_out/r58/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > gpurun_out/r58/trace_summary.txt 2>&1
find gpurun_out/r58 -name "*.db" -delete; find gpurun_out/r58 -name "*.csv" -size +200k -delete
tail -3 gpurun_out/r58/pytest_gpu.txt; tail -3 gpurun_out/r58/smoke.txt; cut -c1-330 gpurun_out/r58/bench.txt; head -9 gpurun_out/r58/trace_summary.txt
echo finished
