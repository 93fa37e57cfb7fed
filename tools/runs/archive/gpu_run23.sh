#!/bin/bash
set -x
mkdir -p gpurun_out/r23
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 300 python tools/prove_bench.py 2 > gpurun_out/r23/prove_serial.txt 2> gpurun_out/r23/prove_serial_err.txt
echo finished
