#!/bin/bash
# round-2 run 21: Y3 of the G1 mixed addition as ONE reduction (fp28::mul_sub2_body): parity + A/B against the same library built with
# -DBZK_G1_FUSED_Y=0 (bazuka_amd/libbzk_ab.so, swapped in for the B leg), same box, alternating
set -x
O=gpurun_out/r02_21
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 40 3 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt | cut -c1-400
cp bazuka_amd/libbzk.so /tmp/libbzk_fused.so
leg() {  # $1 = label
  timeout 300 python bench.py --steps 30 --warmup 5 --no-proofs --no-cpu-baseline --no-overlap --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', 'ms/step', d['ms_per_step'], 'Mpt/s', d['value'], 'accumulate', k['msm_accumulate'], 'avg_launch', d['roofline']['avg_launch_ms'])"
}
for rep in 1 2 3; do
  cp /tmp/libbzk_fused.so bazuka_amd/libbzk.so; leg fused
  cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; leg two_reductions
done | tee $O/ab.txt
cp /tmp/libbzk_fused.so bazuka_amd/libbzk.so
for s in 1 4; do timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done | tee $O/pipe_probe.txt
tail -3 $O/pytest.txt
echo finished
