#!/bin/bash
# run 61: G2 tail kernels without scratch (second operand of an addition read from memory): parity + timings
set -x
mkdir -p gpurun_out/r61
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_worker.py -q -x > gpurun_out/r61/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r61/pytest.txt
tail -3 gpurun_out/r61/pytest.txt
timeout 200 python tools/sweep.py r31 > gpurun_out/r61/sweep.txt 2>&1
cat gpurun_out/r61/sweep.txt
timeout 300 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r61/bench.txt 2> gpurun_out/r61/bench_err.txt
python -c "
import json; d=json.loads(open('gpurun_out/r61/bench.txt').read().strip().splitlines()[-1]); print(d['value'], d['proofs_per_sec'], d['proofs']['gpu_prove_s'], d['proofs']['proofs_per_s_gpu_only'])"
echo finished
