#!/bin/bash
# run 59: clean kernel trace of the MSM headline (no overlapped section) on the final build
set -x
mkdir -p gpurun_out/r59
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CMD="python bench.py --steps 5 --warmup 2 --no-proofs --no-cpu-baseline --no-overlap"
timeout 300 $CMD > gpurun_out/r59/bench_same_cmd.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r59/trace -- $CMD > gpurun_out/r59/trace.log 2>&1
T=$(find gpurun_out/r59/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > gpurun_out/r59/trace_summary.txt 2>&1
find gpurun_out/r59 -name "*.db" -delete; find gpurun_out/r59 -name "*.csv" -size +200k -delete
python -c "
import json; d=json.loads(open('gpurun_out/r59/bench_same_cmd.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
head -12 gpurun_out/r59/trace_summary.txt
echo finished
