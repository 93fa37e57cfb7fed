#!/bin/bash
# round-2 run 2: counting sort of the pairs (atomics) vs rocPRIM radix sort, bucket ordering on 16-bit clamped counts, base
# conversion on the side stream: parity (MSM + groth16 suites), A/B sweep, default bench
set -x
mkdir -p gpurun_out/r02_2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -m gpu -q -x > gpurun_out/r02_2/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_2/pytest.txt
timeout 600 python tools/sweep.py r2csort > gpurun_out/r02_2/sweep.txt 2>&1
timeout 600 python bench.py > gpurun_out/r02_2/bench.txt 2> gpurun_out/r02_2/bench_err.txt
tail -5 gpurun_out/r02_2/pytest.txt; cat gpurun_out/r02_2/sweep.txt; cut -c1-1500 gpurun_out/r02_2/bench.txt
