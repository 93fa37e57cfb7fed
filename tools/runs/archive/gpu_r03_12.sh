#!/bin/bash
# round-3 run 12: b_g1 on its own lane (BZK_PROVE_LANES=4) vs behind l (3): single-proof timeline and 4-slot ceiling; parity of the 4-lane form
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run12; mkdir -p $O
BZK_PROVE_LANES=4 timeout 600 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py -x -q -m gpu > $O/pytest_lanes4.txt 2>&1; echo "rc=$?" >> $O/pytest_lanes4.txt; tail -2 $O/pytest_lanes4.txt
for LN in 3 4 3 4; do
BZK_PROVE_LANES=$LN BZK_TIMING=1 timeout 300 python tools/prove_bench.py 6 > $O/prove_timing_$LN.txt 2>&1; echo "== BZK_PROVE_LANES=$LN"; grep "groth16_prove:" $O/prove_timing_$LN.txt | tail -3 | cut -c60-300; tail -1 $O/prove_timing_$LN.txt | cut -c150-330
BZK_PROVE_LANES=$LN timeout 300 python tools/pipe_probe.py > $O/pipe_probe_$LN.txt 2>&1; tail -1 $O/pipe_probe_$LN.txt
done
