#!/bin/bash
# round-1 run 16: vector-argument field product (no scratch ABI), G2 accumulate at 1 wave/SIMD, 4 producers
set -x
mkdir -p gpurun_out/r16
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r16/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r16/pytest_gpu.txt
BZK_TIMING=1 timeout 600 python bench.py > gpurun_out/r16/bench.txt 2> gpurun_out/r16/bench_err.txt
timeout 400 python tools/sweep.py r16 > gpurun_out/r16/sweep.txt 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r16/trace -- $CMD > gpurun_out/r16/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r16/pmc_fetch -- $CMD > gpurun_out/r16/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r16/pmc_write -- $CMD > gpurun_out/r16/pmc_write.log 2>&1
F=$(find gpurun_out/r16/pmc_fetch -name "*.db" | head -1); W=$(find gpurun_out/r16/pmc_write -name "*.db" | head -1); T=$(find gpurun_out/r16/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > gpurun_out/r16/trace_summary.txt 2>&1
python tools/rocpd_summary.py $F $W > gpurun_out/r16/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $F $W msm_accumulate gpurun_out/r16/pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > gpurun_out/r16/pmc_traffic.log 2>&1
find gpurun_out/r16 -name "*.db" -delete
echo finished
