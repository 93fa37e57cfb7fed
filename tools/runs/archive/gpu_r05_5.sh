#!/bin/bash
# round-5 run 5: the pair-lane G2 tails after the return-address fix (run 4 hung in g2p_add_ni: a long-branch expansion had taken s[30:31]); every command
# under its own short timeout.  Parity first (fixtures, every G2 path), then same-box A/B of the three G2 forms, the per-kernel table of one proof,
# the pipelined rate per form, and what the reduction's existing forms give the stand-alone G1 MSM.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run5; mkdir -p $O
export TMPDIR=/tmp
export SWEEP_CHILD_TIMEOUT=60
( time timeout 90 python -m pytest tests/test_golden_gpu.py -m gpu -q -x ) > $O/pytest_golden.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_golden.txt; tail -4 $O/pytest_golden.txt
if [ $rc -ne 0 ]; then BZK_DEBUG=1 timeout 60 python -m pytest tests/test_golden_gpu.py -m gpu -q -x -k msm_fixtures > $O/debug_launches.txt 2>&1; tail -30 $O/debug_launches.txt | cut -c1-200; echo finished-early; exit 0; fi
( time timeout 420 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_groth16.py tests/test_gpu_fullsize.py tests/test_gpu_mpn_prove.py -m gpu -q -x -k "not 2p24_vs_oracle and not 2p26 and not tree_2p24" --durations=5 ) > $O/pytest_g2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_g2.txt
tail -10 $O/pytest_g2.txt
timeout 300 python tools/sweep.py r5g2 > $O/g2_forms_ab.txt 2>&1
cut -c1-600 $O/g2_forms_ab.txt
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; echo "BZK_G2_PAIR=$1 BZK_G2_PAIR_TAILS=$2"; BZK_G2_PAIR=$1 BZK_G2_PAIR_TAILS=$2 timeout 100 python tools/pipe_probe.py 4 16 2>&1 | tail -1 | cut -c1-400; done > $O/pipe_probe_ab.txt 2>&1
cat $O/pipe_probe_ab.txt
BZK_PROVE_SERIAL=1 timeout 150 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_proof_kernel_table.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -40 $O/serial_proof_kernel_table.txt | cut -c1-150
timeout 200 python tools/sweep.py r5g1tails > $O/g1_tails_forms.txt 2>&1
cut -c1-500 $O/g1_tails_forms.txt
echo finished
