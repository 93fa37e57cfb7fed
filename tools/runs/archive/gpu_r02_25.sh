#!/bin/bash
# round-2 run 25: per-MSM kernel breakdown of one proof (MSMs run one after the other: BZK_PROVE_SERIAL=1) on the current build
set -x
O=gpurun_out/r02_25
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > $O/prove_serial.txt 2> $O/prove_serial_err.txt
grep "serial\|groth16_prove" $O/prove_serial_err.txt | tail -7 | cut -c1-900
tail -1 $O/prove_serial.txt | cut -c1-1500
echo finished
