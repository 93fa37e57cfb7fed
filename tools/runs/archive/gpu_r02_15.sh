#!/bin/bash
# round-2 run 15: is the pipelined proof rate bound by the GPU or by the host side?  N slots proving one pre-built witness in a loop
set -x
O=gpurun_out/r02_15
mkdir -p $O
cd $GRAFT_REPO_ROOT
for s in 1 2 4 6; do timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done | tee $O/pipe_probe.txt
echo finished
