#!/bin/bash
# round-2 run 31: software-pipelined base gathers in msm_accumulate (next base requested inside the current addition, after its last
# product call): parity + A/B against the same sources built with -DBZK_MSM_PREFETCH=0 (bazuka_amd/libbzk_ab.so), same box, alternating
set -x
O=gpurun_out/r02_31
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 30 9 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
cp bazuka_amd/libbzk.so /tmp/libbzk_new.so
leg() {
  timeout 300 python bench.py --steps 30 --warmup 5 --no-proofs --no-cpu-baseline --no-overlap --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', 'g1 ms/step', d['ms_per_step'], 'Mpt/s', d['value'], 'accumulate', k['msm_accumulate'])"
  timeout 200 python tools/sweep.py child g2 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'g2 ms', d['ms'], 'accumulate', d['prof']['msm_accumulate'])"
  timeout 200 python tools/sweep.py child g1 24 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'g1 2^24 ms', d['ms'], 'accumulate', d['prof']['msm_accumulate'])"
}
for rep in 1 2; do
  cp /tmp/libbzk_new.so bazuka_amd/libbzk.so; leg prefetch
  cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; leg plain
done | tee $O/ab.txt
for which in new ab; do
  if [ $which = new ]; then cp /tmp/libbzk_new.so bazuka_amd/libbzk.so; else cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; fi
  for s in 1 4; do echo -n "$which "; timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done
done | tee $O/pipe_probe.txt
cp /tmp/libbzk_new.so bazuka_amd/libbzk.so
tail -3 $O/pytest.txt
echo finished
