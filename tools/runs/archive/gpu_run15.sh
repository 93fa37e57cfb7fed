#!/bin/bash
# round-1 run 15: full GPU test suite, bench with lane-parallel prove + timing, G2 occupancy A/B, rocprofv3 trace + PMC passes
set -x
mkdir -p gpurun_out/r15
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r15/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r15/pytest_gpu.txt
BZK_TIMING=1 timeout 600 python bench.py > gpurun_out/r15/bench.txt 2> gpurun_out/r15/bench_err.txt
timeout 400 python tools/sweep.py g2occ > gpurun_out/r15/g2occ.txt 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r15/trace -- $CMD > gpurun_out/r15/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r15/pmc_fetch -- $CMD > gpurun_out/r15/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r15/pmc_write -- $CMD > gpurun_out/r15/pmc_write.log 2>&1
F=$(find gpurun_out/r15/pmc_fetch -name "*.db" | head -1); W=$(find gpurun_out/r15/pmc_write -name "*.db" | head -1); T=$(find gpurun_out/r15/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > gpurun_out/r15/trace_summary.txt 2>&1
python tools/rocpd_summary.py $F $W > gpurun_out/r15/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $F $W msm_accumulate gpurun_out/r15/pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > gpurun_out/r15/pmc_traffic.log 2>&1
find gpurun_out/r15 -name "*.db" -size +20M -delete
echo finished
