#!/bin/bash
# round-2 run 48: the kernel trace of the bench command at its DEFAULT step counts (20 + 3 warm-up + 5 instrumented), so that the trace's
# average msm_accumulate duration and the HIP-event figure the same process prints (roofline.avg_launch_ms) can be compared directly
O=gpurun_out/r02_48
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CMD="python bench.py --no-proofs --no-cpu-baseline --no-others --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +300k -delete
head -12 $O/trace_summary.txt
grep -o '"avg_launch_ms": [0-9.]*\|"ms_per_step": [0-9.]*\|"value": [0-9.]*' $O/trace.log | head -4
echo finished
