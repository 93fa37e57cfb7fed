#!/bin/bash
# run 63: per-MSM kernel breakdown of ONE proof of the final build, MSMs run serially (BZK_PROVE_SERIAL=1)
set -x
mkdir -p gpurun_out/r63
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > gpurun_out/r63/prove_serial.txt 2> gpurun_out/r63/prove_serial_err.txt
grep "serial " gpurun_out/r63/prove_serial_err.txt | tail -4 > gpurun_out/r63/serial_last_proof.txt
grep "groth16_prove:" gpurun_out/r63/prove_serial_err.txt | tail -2 >> gpurun_out/r63/serial_last_proof.txt
rm -f gpurun_out/r63/prove_serial_err.txt
cut -c1-900 gpurun_out/r63/serial_last_proof.txt
echo finished
