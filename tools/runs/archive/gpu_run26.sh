#!/bin/bash
# round-1 run 26: NTT on the reduced-radix Fr (DIT in LDS), priority stream for the G2 lane
set -x
mkdir -p gpurun_out/r26
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -x -q > gpurun_out/r26/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r26/pytest.txt
BZK_NTT_BMAX=4 timeout 300 python -m pytest tests/test_gpu_poseidon_ntt.py -x -q -k "ntt_vs_oracle and not 21 and not 22 and not 16 and not 17" > gpurun_out/r26/pytest_ntt_bmax4.txt 2>&1; echo "rc=$?" >> gpurun_out/r26/pytest_ntt_bmax4.txt
timeout 600 python tools/sweep.py r26 > gpurun_out/r26/sweep.txt 2>&1
BZK_TIMING=1 timeout 600 python bench.py --steps 10 --no-cpu-baseline > gpurun_out/r26/bench.txt 2> gpurun_out/r26/bench_err.txt
echo finished
