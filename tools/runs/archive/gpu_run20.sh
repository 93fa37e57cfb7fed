#!/bin/bash
# round-1 run 20: full test suite on the current build, bench, serial-prove kernel breakdown, SQ counters of the accumulate kernel
set -x
mkdir -p gpurun_out/r20
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r20/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r20/pytest_gpu.txt
BZK_TIMING=1 timeout 600 python bench.py > gpurun_out/r20/bench.txt 2> gpurun_out/r20/bench_err.txt
BZK_PROVE_SERIAL=1 timeout 300 python tools/prove_bench.py 3 > gpurun_out/r20/prove_serial.txt 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/r20/pmc_sq1 -- $CMD > gpurun_out/r20/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d gpurun_out/r20/pmc_sq2 -- $CMD > gpurun_out/r20/pmc_sq2.log 2>&1
for d in pmc_sq1 pmc_sq2; do F=$(find gpurun_out/r20/$d -name "*.db" | head -1); python tools/rocpd_summary.py $F > gpurun_out/r20/${d}_summary.txt 2>&1; done
find gpurun_out/r20 -name "*.db" -delete
echo finished
