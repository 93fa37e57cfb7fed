#!/bin/bash
# run 56: dedicated Fp28 squaring (301 instead of 392 mads per square): parity + kernel times
set -x
mkdir -p gpurun_out/r56
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -q -x > gpurun_out/r56/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r56/pytest.txt
tail -3 gpurun_out/r56/pytest.txt
timeout 300 python tools/sweep.py r36 > gpurun_out/r56/sweep.txt 2>&1
cat gpurun_out/r56/sweep.txt
timeout 300 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r56/bench.txt 2> gpurun_out/r56/bench_err.txt
python -c "
import json; d=json.loads(open('gpurun_out/r56/bench.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['proofs_per_sec'], d['proofs']['gpu_prove_s'], d.get('two_msms_in_flight',{}).get('value'))"
echo finished
