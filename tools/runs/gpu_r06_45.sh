#!/bin/bash
# round-6 run 45: ten minutes of the whole fuzzer on the shipped tree (two seeds)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run45; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tests/tools/fuzz_gpu.py 300 4501 > $O/fuzz_a.txt 2>&1; echo "rc=$?" >> $O/fuzz_a.txt; tail -2 $O/fuzz_a.txt | cut -c1-600
AMD_DIRECT_DISPATCH=0 timeout 400 python tests/tools/fuzz_gpu.py 300 4502 > $O/fuzz_b.txt 2>&1; echo "rc=$?" >> $O/fuzz_b.txt; tail -2 $O/fuzz_b.txt | cut -c1-600
echo finished
