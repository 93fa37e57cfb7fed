#!/bin/bash
# round-3 run 19: l + r b_g1 as one MSM (BZK_PROVE_MERGE_LB1): parity of proofs, then A/B (0|1 alternating) of the single-proof
# timeline and of the 4-slot ceiling (same witness, no producers)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_worker.py tests/test_gpu_mg.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for V in 0 1 0 1; do
echo "== BZK_PROVE_MERGE_LB1=$V"
BZK_PROVE_MERGE_LB1=$V BZK_TIMING=1 timeout 300 python tools/prove_bench.py 6 > $O/prove_timing_$V.txt 2>&1; grep "groth16_prove:" $O/prove_timing_$V.txt | tail -4 | cut -c60-330; tail -1 $O/prove_timing_$V.txt | cut -c150-330
BZK_PROVE_MERGE_LB1=$V timeout 300 python tools/pipe_probe.py > $O/pipe_probe_$V.txt 2>&1; tail -1 $O/pipe_probe_$V.txt
done 2>&1 | grep -v "^+" | tee $O/ab.txt
