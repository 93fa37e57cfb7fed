#!/bin/bash
# round-2 run 22: window-in-value sort keys (bucket-within-window key = one radix pass fewer): parity + A/B (BZK_MSM_NO_WIV=1 = old keys)
set -x
O=gpurun_out/r02_22
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not tree_2p24 and not ntt_2p24" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 40 4 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt | cut -c1-400
leg() {
  timeout 300 python bench.py --steps 30 --warmup 5 --no-proofs --no-cpu-baseline --no-overlap --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', 'ms/step', d['ms_per_step'], 'Mpt/s', d['value'], {x:k[x] for x in ('msm_sort_pairs','msm_offsets','msm_accumulate','msm_digits')})"
}
for rep in 1 2 3; do
  BZK_MSM_NO_WIV=0 leg wiv
  BZK_MSM_NO_WIV=1 leg plain_keys
done | tee $O/ab.txt
for w in 0 1; do for s in 1 4; do echo -n "NO_WIV=$w "; BZK_MSM_NO_WIV=$w timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done; done | tee $O/pipe_probe.txt
for lg in 22 24; do for w in 0 1; do echo -n "NO_WIV=$w 2^$lg: "; BZK_MSM_NO_WIV=$w timeout 200 python tools/sweep.py child g1 $lg 2>/dev/null | tail -1 | cut -c1-500; done; done | tee $O/sweep24.txt
tail -3 $O/pytest.txt
echo finished
