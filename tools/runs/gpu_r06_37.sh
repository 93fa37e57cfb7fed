#!/bin/bash
# round-6 run 37: where the prover side's host CPU goes (tools/host_cpu_probe.py): blocking waits against spinning, 4 slots and 1
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run37; mkdir -p $O
export TMPDIR=/tmp
for b in 1 0; do for sl in 4 1; do BZK_SYNC_BLOCKING=$b timeout 300 python tools/host_cpu_probe.py $sl 24 2>&1 | grep '^{'; done; done > $O/host_cpu.txt
cat $O/host_cpu.txt
echo finished
