#!/bin/bash
# round-4 run 17: per-kernel table of serial proofs on the committed build (endomorphism form on for the prover's MSMs)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run17; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BZK_PROVE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial_trace.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_trace_summary.txt 2>&1
rm -rf $O/serial_trace
tail -3 $O/serial_trace.log; head -50 $O/serial_trace_summary.txt | cut -c1-150
echo finished
