#!/bin/bash
# round-6 run 1: (A) the new device-fill parity tests; (B) how much independent MSMs overlap, 1..4 in flight, and whether the number of
# hardware queues matters; (C) the kernel timeline of two MSMs in flight; (D) the pipelined proofs ceiling (tools/pipe_probe.py) against GPU_MAX_HW_QUEUES + its timeline
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_defer.py -m gpu -q -p pytest_timeout --timeout=420 --durations=8 ) > $O/pytest_defer.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_defer.txt
tail -14 $O/pytest_defer.txt | cut -c1-200
for Q in default 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  echo "== GPU_MAX_HW_QUEUES=$Q" >> $O/overlap.txt
  timeout 200 python tools/overlap_probe.py 1,2,3,4 16 20 >> $O/overlap.txt 2>$O/overlap_err_$Q.txt
done
unset GPU_MAX_HW_QUEUES
echo "== throughput forms (flag 4)" >> $O/overlap.txt
timeout 200 python tools/overlap_probe.py 1,2,4 16 20 4 >> $O/overlap.txt 2>>$O/overlap_err.txt
cat $O/overlap.txt
timeout 200 rocprofv3 --kernel-trace -d $O/trace_k2 -- python tools/overlap_probe.py 2 12 20 > $O/trace_k2.log 2>&1
T=$(find $O/trace_k2 -name "*.db" | head -1); python tools/trace_timeline.py $T auto 10 140 > $O/timeline_k2.txt 2>&1
head -60 $O/timeline_k2.txt | cut -c1-160
for Q in default 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  echo "== GPU_MAX_HW_QUEUES=$Q" >> $O/pipe.txt
  timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_$Q.txt
done
unset GPU_MAX_HW_QUEUES
cat $O/pipe.txt | cut -c1-400
timeout 300 rocprofv3 --kernel-trace -d $O/trace_pipe -- python tools/pipe_probe.py 4 12 > $O/trace_pipe.log 2>&1
T=$(find $O/trace_pipe -name "*.db" | head -1); python tools/trace_timeline.py $T auto 32 400 > $O/timeline_pipe.txt 2>&1
python tools/trace_busy.py $T 0.5 > $O/pipe_busy.txt 2>&1
head -40 $O/timeline_pipe.txt | cut -c1-160
cat $O/pipe_busy.txt | head -30
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo finished
