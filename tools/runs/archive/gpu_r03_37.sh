#!/bin/bash
# round-3 run 37: rocprofv3 kernel trace of the bench command at its DEFAULT step counts: msm_accumulate average of the trace vs the
# HIP-event figure the same process reports (roofline.avg_launch_ms), bounded
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run37; mkdir -p $O
export TMPDIR=/tmp
CMD="python bench.py --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "rc=$?"
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -8 $O/trace_summary.txt | cut -c1-160
grep '"metric"' $O/trace.log | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('bench line of the traced process: value', d['value'], 'ms_per_step', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], 'steps', d['steps'], 'warmup', d['warmup'])"
