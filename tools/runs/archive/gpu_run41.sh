#!/bin/bash
# round-1 run 41: final-state evidence: full GPU suite, default bench, kernel trace of the bench incl. the proof section
set -x
mkdir -p gpurun_out/r41
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r41/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r41/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r41/bench.txt 2> gpurun_out/r41/bench_err.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r41/smoke.txt 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r41/trace -- $CMD > gpurun_out/r41/trace.log 2>&1
T=$(find gpurun_out/r41/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > gpurun_out/r41/trace_summary.txt 2>&1
CMD2="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-proofs"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r41/trace_msm -- $CMD2 > gpurun_out/r41/trace_msm.log 2>&1
T2=$(find gpurun_out/r41/trace_msm -name "*.db" | head -1); python tools/rocpd_summary.py $T2 > gpurun_out/r41/trace_msm_summary.txt 2>&1
find gpurun_out/r41 -name "*.db" -delete
timeout 300 python tools/tree_bench.py > gpurun_out/r41/tree_bench.txt 2>&1
echo finished
