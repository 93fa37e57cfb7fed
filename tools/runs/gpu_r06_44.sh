#!/bin/bash
# round-6 run 44: with the launches through the runtime's worker threads (AMD_DIRECT_DISPATCH=0): prover slots 3 .. 6, lanes 3 / 4, hardware queues 16 / 24 / 32 (pipe_probe: same witness, no producers)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run44; mkdir -p $O
export TMPDIR=/tmp
export AMD_DIRECT_DISPATCH=0 BZK_SYNC_BLOCKING=1
( for rep in 1 2; do
for sl in 4 3 5 6; do echo "## slots $sl"; timeout 300 python tools/pipe_probe.py $sl 24 2>/dev/null | tail -1; done
echo "## slots 4 lanes 4"; BZK_PROVE_LANES=4 timeout 300 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1
echo "## slots 4 queues 32"; GPU_MAX_HW_QUEUES=32 timeout 300 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1
echo "## slots 6 queues 32"; GPU_MAX_HW_QUEUES=32 timeout 300 python tools/pipe_probe.py 6 24 2>/dev/null | tail -1
done ) > $O/slots.txt 2>&1
cat $O/slots.txt
echo finished
