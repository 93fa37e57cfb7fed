#!/bin/bash
# round-2 run 27: production sizes on the final arithmetic (256-tx Update circuit, 2^24 domain, pairing check; one production block)
set -x
O=gpurun_out/r02_27
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python tests/tools/prove_production.py 4 3 0 > $O/production_256tx.txt 2> $O/production_256tx_err.txt; tail -2 $O/production_256tx_err.txt
timeout 300 python tests/tools/prove_block.py > $O/production_block.txt 2>&1
tail -1 $O/production_256tx.txt | cut -c1-1400; tail -3 $O/production_block.txt | cut -c1-1200
echo finished
