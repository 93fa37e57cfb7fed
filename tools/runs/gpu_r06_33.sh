#!/bin/bash
# round-6 run 33: soak of the new paths - the fuzzer restricted to window ranges in flight + giant buckets (both curves) for 6 minutes, then the whole fuzzer for 2
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run33; mkdir -p $O
export TMPDIR=/tmp
FUZZ_KINDS=ranges timeout 600 python tests/tools/fuzz_gpu.py 360 3301 > $O/fuzz_ranges.txt 2>&1; echo "rc=$?" >> $O/fuzz_ranges.txt; tail -3 $O/fuzz_ranges.txt | cut -c1-600
timeout 300 python tests/tools/fuzz_gpu.py 120 3302 > $O/fuzz_all.txt 2>&1; echo "rc=$?" >> $O/fuzz_all.txt; tail -2 $O/fuzz_all.txt | cut -c1-600
echo finished
