#!/bin/bash
# round-5 run 1: (a) the parity holes VERDICT r4 names - 2^24 and 2^26 G1 MSM vs the oracle incl. the 8-way window-sharded fold, the single 1024-tx circuit;
# (b) the per-kernel table of ONE proof with the shipped defaults (endomorphism form in the prover's calls); (c) what the existing switches give the stand-alone MSMs.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run1; mkdir -p $O
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/memory.max; free -g | head -2
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_production.py -m gpu -q -x -k "2p24_vs_oracle or 2p26 or single_1024tx" --durations=5 ) > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt
tail -12 $O/pytest_new.txt
BZK_PROVE_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_proof_kernel_table.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -45 $O/serial_proof_kernel_table.txt | cut -c1-150
timeout 600 python tools/sweep.py r5knobs > $O/knobs.txt 2>&1
cat $O/knobs.txt | cut -c1-600
echo finished
