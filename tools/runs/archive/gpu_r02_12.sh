#!/bin/bash
# round-2 run 12: timing of the device-resident MPN account state at the production depth (L = 15, T = 3)
set -x
O=gpurun_out/r02_12
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/mpn_tree_bench.py > $O/mpn_tree_bench.txt 2> $O/mpn_tree_bench_err.txt
cat $O/mpn_tree_bench.txt; tail -3 $O/mpn_tree_bench_err.txt
echo finished
