#!/bin/bash
# run 49: built-in work-model window pick for the witness MSMs vs the plain pick (env -1) and fixed choices; parity tests
set -x
mkdir -p gpurun_out/r49
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/r49/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r49/pytest.txt
tail -3 gpurun_out/r49/pytest.txt
for cfg in "-1 -1" "0 0" "14 13" "0 0" "-1 -1" "13 13" "15 14"; do
  set -- $cfg
  echo "## BZK_MSM_C_WIT_G1=$1 BZK_MSM_C_WIT_G2=$2   (0 = built-in model, -1 = plain pick)" >> gpurun_out/r49/ab.txt
  BZK_MSM_C_WIT_G1=$1 BZK_MSM_C_WIT_G2=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>> gpurun_out/r49/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proofs']
print(json.dumps({k:p.get(k) for k in ('gpu_prove_s','proofs_per_s_gpu_only','proofs_per_s_pipelined','producer_synth_s_mean_under_load')}))" >> gpurun_out/r49/ab.txt
done
cat gpurun_out/r49/ab.txt
timeout 300 python tests/tools/prove_production.py 4 3 0 > gpurun_out/r49/production_256tx.txt 2>&1
cut -c1-600 gpurun_out/r49/production_256tx.txt | tail -2
echo finished
