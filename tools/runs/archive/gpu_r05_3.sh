#!/bin/bash
# round-5 run 3: the G2 tails on pairs of lanes (msm_g2pair_tails.cuh): parity of every G2 path through the C ABI, same-box A/B against the one-lane tails
# (stand-alone MSM, pipelined proofs), the per-kernel table of one proof; the 1024-tx circuit with the host work builder
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_groth16.py tests/test_gpu_fullsize.py tests/test_gpu_mpn_prove.py -m gpu -q -x -k "not 2p24_vs_oracle and not 2p26" --durations=5 ) > $O/pytest_g2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_g2.txt
tail -12 $O/pytest_g2.txt
timeout 600 python tools/sweep.py r5tails > $O/tails_ab.txt 2>&1
cut -c1-700 $O/tails_ab.txt
for rep in 1 2; do for tails in 0 1; do echo "BZK_G2_PAIR_TAILS=$tails"; BZK_G2_PAIR_TAILS=$tails timeout 200 python tools/pipe_probe.py 4 16 2>&1 | tail -1 | cut -c1-400; done; done > $O/pipe_probe_ab.txt 2>&1
cat $O/pipe_probe_ab.txt
BZK_PROVE_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_proof_kernel_table.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -32 $O/serial_proof_kernel_table.txt | cut -c1-150
( time timeout 900 python -m pytest tests/test_gpu_production.py -m gpu -q -x -k "single_1024tx or withdraw_15_3_3" --durations=5 ) > $O/pytest_rest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.txt
tail -12 $O/pytest_rest.txt
echo finished
