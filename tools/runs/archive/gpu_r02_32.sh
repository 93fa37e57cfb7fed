#!/bin/bash
# round-2 run 32: msm_accumulate<G1> compiled for 3 waves per SIMD (168 registers, 54 spilled) against 2 (194 registers, none), same box
set -x
O=gpurun_out/r02_32
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp bazuka_amd/libbzk.so /tmp/libbzk_new.so
leg() {
  timeout 300 python bench.py --steps 30 --warmup 5 --no-proofs --no-cpu-baseline --no-overlap --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', 'g1 ms/step', d['ms_per_step'], 'Mpt/s', d['value'], 'accumulate', k['msm_accumulate'])"
}
for rep in 1 2; do
  cp /tmp/libbzk_new.so bazuka_amd/libbzk.so; leg occ2
  cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; leg occ3
done | tee $O/ab.txt
cp /tmp/libbzk_new.so bazuka_amd/libbzk.so
echo finished
