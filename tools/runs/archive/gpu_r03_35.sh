#!/bin/bash
# round-3 run 35: production-size witnesses and proofs after the host witness fast path (bounded)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run35; mkdir -p $O
timeout 200 python tests/tools/prove_block.py > $O/production_block.txt 2>&1; echo "rc=$?" >> $O/production_block.txt; tail -2 $O/production_block.txt | cut -c1-1000
timeout 150 python tests/tools/prove_production.py 4 3 0 1 > $O/production_256tx_device_builder.txt 2>&1; echo "rc=$?"; tail -1 $O/production_256tx_device_builder.txt | cut -c1-500
