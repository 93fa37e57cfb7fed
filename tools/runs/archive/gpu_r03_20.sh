#!/bin/bash
# round-3 run 20: where the main stream of a proof spends its 20 ms (device-side marks), alone and beside the lanes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run20; mkdir -p $O
BZK_TIMING=1 timeout 300 python tools/prove_bench.py 6 > $O/prove_timing.txt 2>&1; grep "groth16_prove" $O/prove_timing.txt | tail -8 | cut -c1-330
BZK_PROVE_SERIAL=1 BZK_TIMING=1 timeout 300 python tools/prove_bench.py 4 > $O/prove_timing_serial.txt 2>&1; grep "groth16_prove" $O/prove_timing_serial.txt | tail -6 | cut -c1-330
