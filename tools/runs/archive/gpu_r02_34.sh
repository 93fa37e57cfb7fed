#!/bin/bash
# round-2 run 34: where do the ~7 % between the pipelined bench figure (host producers) and the GPU-side probe go?  producers x threads sweep
set -x
O=gpurun_out/r02_34
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "6 16" "4 16" "3 24" "8 8" "4 32" "6 16"; do
  set -- $cfg
  echo -n "producers=$1 threads=$2 "
  BZK_BENCH_PRODUCERS=$1 BZK_BENCH_PROD_THREADS=$2 timeout 200 python bench.py --steps 3 --warmup 1 --no-others --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proofs']
print('pipelined', p.get('proofs_per_s_pipelined'), 'synth_under_load', p.get('producer_synth_s_mean_under_load'), 'gpu_prove_s', p.get('gpu_prove_s'))"
done | tee $O/sweep.txt
for s in 1 4; do timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done | tee $O/pipe_probe.txt
echo finished
