#!/bin/bash
# round-3 run 10: where a single proof's wall time goes (BZK_TIMING: main chain vs lanes), serial per-MSM kernel breakdown, GPU-side ceiling
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run10; mkdir -p $O
BZK_TIMING=1 timeout 300 python tools/prove_bench.py 4 > $O/prove_timing.txt 2>&1; grep "groth16_prove:" $O/prove_timing.txt | tail -4
BZK_PROVE_SERIAL=1 BZK_TIMING=1 timeout 300 python tools/prove_bench.py 3 > $O/prove_serial.txt 2>&1; grep "serial\|groth16_prove:" $O/prove_serial.txt | tail -6 | cut -c1-1200
timeout 300 python tools/pipe_probe.py > $O/pipe_probe.txt 2>&1; tail -4 $O/pipe_probe.txt
