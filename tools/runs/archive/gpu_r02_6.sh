#!/bin/bash
# round-2 run 6: LDS two-pass partition of the pairs (BZK_MSM_PSORT, default on) vs the rocPRIM radix sort: parity + A/B + bench
set -x
mkdir -p gpurun_out/r02_6
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_golden_gpu.py -m gpu -q -x > gpurun_out/r02_6/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_6/pytest.txt
timeout 600 python tools/sweep.py r2psort > gpurun_out/r02_6/sweep.txt 2>&1
timeout 300 python bench.py --no-others --no-cpu-baseline > gpurun_out/r02_6/bench.txt 2>/dev/null
tail -4 gpurun_out/r02_6/pytest.txt; cat gpurun_out/r02_6/sweep.txt; cut -c1-2200 gpurun_out/r02_6/bench.txt
