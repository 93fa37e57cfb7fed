#!/bin/bash
# round-3 run 33: rocprofv3 kernel trace of full proofs on the last build (tools/prove_bench.py: 6 single proofs + 6 pipelined + 1), bounded
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run33; mkdir -p $O
export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/prove_bench.py 6 > $O/trace.log 2>&1; echo "rc=$?"
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -40 $O/trace_summary.txt | cut -c1-160
tail -1 $O/trace.log | cut -c1-400
