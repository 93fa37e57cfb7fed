#!/bin/bash
# round-2 run 40: does the pinned pool keep allocating while witness producers run beside a prover (hipHostMalloc serialises with GPU work)?
O=gpurun_out/r02_40
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{
BZK_POOL_DEBUG=1 timeout 90 python tools/pipe_probe.py 1 24 8 8 > $O/probe.txt 2> $O/probe_err.txt
tail -1 $O/probe.txt
echo "hipHostMalloc calls in all: $(grep -c 'pool: hipHostMalloc' $O/probe_err.txt)"
grep 'pool: hipHostMalloc' $O/probe_err.txt | awk '{print $4}' | sort -n | uniq -c | sort -rn | head -12
grep 'pool: hipHostMalloc' $O/probe_err.txt | tail -3
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
