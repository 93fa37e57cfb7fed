#!/bin/bash
# run 48: window size of the witness MSMs (BZK_F_DEDUP) per curve, judged by pipelined proofs/s (work, not latency)
set -x
mkdir -p gpurun_out/r48
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "0 0" "0 14" "0 13" "15 14" "14 13" "0 0" "14 14" "0 12"; do
  set -- $cfg
  echo "## BZK_MSM_C_WIT_G1=$1 BZK_MSM_C_WIT_G2=$2" >> gpurun_out/r48/ab.txt
  BZK_MSM_C_WIT_G1=$1 BZK_MSM_C_WIT_G2=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>> gpurun_out/r48/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proofs']
print(json.dumps({k:p.get(k) for k in ('gpu_prove_s','proofs_per_s_gpu_only','proofs_per_s_pipelined','producer_synth_s_mean_under_load')}))" >> gpurun_out/r48/ab.txt
done
cat gpurun_out/r48/ab.txt
echo finished
