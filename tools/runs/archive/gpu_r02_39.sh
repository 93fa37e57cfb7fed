#!/bin/bash
# round-2 run 39: what do witness producers take from the prover (run 37: 54 -> 34 proofs/s with producers running flat out beside it)?
# the 4-slot probe under synthetic host load: N spinning processes (CPU only), N memory-streaming processes (bandwidth only) - every load
# process ends by itself; and the prover's own phase timings (BZK_TIMING) of a 1-slot probe with / without background producers
O=gpurun_out/r02_39
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{
nproc; lscpu | grep -i "numa node\|socket\|model name\|thread" | head -8
probe() { timeout 90 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1; }
echo -n "no load: "; probe
for n in 64 200; do timeout 60 python tools/host_load.py cpu $n 28 > /dev/null 2>&1 & sleep 2; echo -n "cpu x$n: "; probe; wait; done
for n in 8 32; do timeout 60 python tools/host_load.py mem $n 28 > /dev/null 2>&1 & sleep 3; echo -n "mem x$n: "; probe; wait; done
echo -n "no load: "; probe
echo "## BZK_TIMING of a 1-slot probe, last 3 proofs: without / with 8 x 8 background producers"
BZK_TIMING=1 timeout 90 python tools/pipe_probe.py 1 12 0 8 2>&1 | grep "groth16_prove:\|proofs_per_s" | tail -4
BZK_TIMING=1 timeout 90 python tools/pipe_probe.py 1 12 8 8 2>&1 | grep "groth16_prove:\|proofs_per_s" | tail -4
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
