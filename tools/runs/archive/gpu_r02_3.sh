#!/bin/bash
# round-2 run 3: device-resident MPN account state (bzk_mpn_tree_*) parity + the ADVICE tests + bench with other_configs
set -x
mkdir -p gpurun_out/r02_3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mpn_tree.py tests/test_gpu_tree4.py tests/test_gpu_groth16.py tests/test_gpu_msm.py -m gpu -q -x --durations=8 > gpurun_out/r02_3/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_3/pytest.txt
timeout 600 python bench.py > gpurun_out/r02_3/bench.txt 2> gpurun_out/r02_3/bench_err.txt
tail -25 gpurun_out/r02_3/pytest.txt; cat gpurun_out/r02_3/bench.txt; tail -5 gpurun_out/r02_3/bench_err.txt
