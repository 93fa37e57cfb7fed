#!/bin/bash
# round-4 run 23: do the quad-lane (latency) forms of the G1 tails cost the PIPELINED prover throughput?  A/B on the proofs leg:
# BZK_MSM_QUAD_L2 / BZK_MSM_QUAD_TREE = 1 (default: quads of lanes, ~3x the instructions, shorter chain) vs 0 (one lane per chunk / point)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run23; mkdir -p $O
CMD="python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline --no-production"
for rep in 1 2; do
for cfg in "1 1" "0 1" "0 0"; do
  set -- $cfg
  echo "== rep $rep BZK_MSM_QUAD_L2=$1 BZK_MSM_QUAD_TREE=$2" >> $O/ab.txt
  BZK_BENCH_TWO_PROCS=0 BZK_MSM_QUAD_L2=$1 BZK_MSM_QUAD_TREE=$2 timeout 300 $CMD 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['proofs']
print({k:p.get(k) for k in ('gpu_prove_s','proofs_per_s_serial','proofs_per_s_pipelined','proofs_per_s_ring')}, d['value'])" >> $O/ab.txt
done; done
cat $O/ab.txt
echo finished
