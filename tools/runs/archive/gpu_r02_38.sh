#!/bin/bash
# round-2 run 38: what do idle-running witness producers take from the prover (run 37: 54 -> 34 proofs/s)?  the same probe under synthetic host
# load: N spinning processes (CPU only) and N memory-streaming processes (bandwidth only)
set -x
O=gpurun_out/r02_38
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc; lscpu | grep -i "numa\|socket\|model name" | head -8
probe() { timeout 100 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1; }
echo -n "no load: "; probe
for n in 64 192; do python tools/host_load.py cpu $n 40 & L=$!; sleep 2; echo -n "cpu x$n: "; probe; kill $L; wait $L 2>/dev/null; done
for n in 8 32; do python tools/host_load.py mem $n 40 & L=$!; sleep 3; echo -n "mem x$n: "; probe; kill $L; wait $L 2>/dev/null; done
echo -n "no load: "; probe
echo finished
