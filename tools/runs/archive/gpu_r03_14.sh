#!/bin/bash
# round-3 run 14: BASELINE configs[2] at face value on the round-3 build: ONE 1024-tx Update circuit (57.8 M constraints, 2^26 domain),
# witness through the device builder, pairing-verified
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run14; mkdir -p $O
timeout 900 python tests/tools/prove_production.py 5 3 0 1 > $O/production_1024tx.txt 2>&1; echo "rc=$?" >> $O/production_1024tx.txt
tail -3 $O/production_1024tx.txt | cut -c1-1500
