#!/bin/bash
# round-3 run 11: stream priorities of a proof's lanes (BZK_PRIO = g2 | main | none): single-proof timeline and 4-slot ceiling
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run11; mkdir -p $O
for P in g2 main none; do
BZK_PRIO=$P BZK_TIMING=1 timeout 300 python tools/prove_bench.py 5 > $O/prove_timing_$P.txt 2>&1; echo "== BZK_PRIO=$P"; grep "groth16_prove:" $O/prove_timing_$P.txt | tail -3; tail -1 $O/prove_timing_$P.txt | cut -c1-400
BZK_PRIO=$P timeout 300 python tools/pipe_probe.py > $O/pipe_probe_$P.txt 2>&1; tail -2 $O/pipe_probe_$P.txt
done
