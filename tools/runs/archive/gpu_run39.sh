#!/bin/bash
# round-1 run 39: cooperative (8 lanes per node) width-5 Poseidon for latency-bound launches
set -x
mkdir -p gpurun_out/r39
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_tree4.py tests/test_golden_gpu.py tests/test_gpu_mpn_prove.py -x -q > gpurun_out/r39/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r39/pytest.txt
timeout 300 python tools/fuzz_gpu.py 40 7 > gpurun_out/r39/fuzz.txt 2>&1
for nc in 0 1; do
  BZK_NO_COOP=$nc timeout 300 python tools/sweep.py r39 > gpurun_out/r39/sweep_nocoop$nc.txt 2>&1
  BZK_NO_COOP=$nc timeout 300 python tools/tree_bench.py > gpurun_out/r39/tree_bench_nocoop$nc.txt 2>&1
done
echo finished
