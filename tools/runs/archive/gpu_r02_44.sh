#!/bin/bash
# round-2 run 44: are witness producers harmful as THREADS OF THE PROVER'S PROCESS or as host load in general?  the 4-slot probe with 8 x 8
# producers (output discarded) inside the process and in a separate process that sees no GPU
O=gpurun_out/r02_44
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{
timeout 120 python tools/pipe_probe.py 4 24 0 8 0 2>/dev/null | tail -1
timeout 120 python tools/pipe_probe.py 4 24 8 8 0 2>/dev/null | tail -1
timeout 120 python tools/pipe_probe.py 4 24 8 8 1 2>/dev/null | tail -1
timeout 120 python tools/pipe_probe.py 4 24 16 8 1 2>/dev/null | tail -1
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
