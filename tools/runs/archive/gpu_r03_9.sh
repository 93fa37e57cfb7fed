#!/bin/bash
# round-3 run 9: differential fuzzing of the C ABI incl. the round-3 entry points (resident bases, device groups, fused h chain,
# general state compress), soak of the proof path with 4 slots over ONE shared CRS (every proof verified), production block
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run9; mkdir -p $O
timeout 400 python tests/tools/fuzz_gpu.py 150 3 > $O/fuzz.txt 2>&1; echo "rc=$?" >> $O/fuzz.txt; tail -3 $O/fuzz.txt
timeout 400 python tests/tools/fuzz_gpu.py 100 11 > $O/fuzz2.txt 2>&1; echo "rc=$?" >> $O/fuzz2.txt; tail -3 $O/fuzz2.txt
timeout 600 python tests/tools/soak.py 800 4 > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt; tail -4 $O/soak.txt
timeout 600 python tests/tools/prove_block.py > $O/production_block.txt 2>&1; echo "rc=$?" >> $O/production_block.txt; tail -2 $O/production_block.txt | cut -c1-900
