#!/bin/bash
# round-6 run 2: (A) device-fill parity tests (run 1's pytest line double-registered the timeout plugin) + the G2 parity tests on the LDS-staged pair accumulation;
# (B) same-box A/B of that kernel against the round-5 form; (C) the saturating kernels of an MSM on a lowest-priority stream (BZK_MSM_HEAVY_PRIO): MSMs in flight and
# the pipelined proofs ceiling, with 4 and 16 hardware queues
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_defer.py -m gpu -q --timeout=420 --durations=8 ) > $O/pytest_defer.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_defer.txt
tail -14 $O/pytest_defer.txt | cut -c1-200
( time timeout 900 python -m pytest tests -m gpu -q --timeout=420 -k "g2 or G2 or endo or mpn_prove" --durations=5 ) > $O/pytest_g2.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_g2.txt
tail -10 $O/pytest_g2.txt | cut -c1-200
timeout 600 python tools/sweep.py r6g2lds > $O/g2_lds_ab.txt 2>&1
cat $O/g2_lds_ab.txt | cut -c1-420
for Q in default 16; do
  for H in 0 1; do
    if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
    export BZK_MSM_HEAVY_PRIO=$H
    echo "== GPU_MAX_HW_QUEUES=$Q BZK_MSM_HEAVY_PRIO=$H" >> $O/overlap.txt
    timeout 200 python tools/overlap_probe.py 1,2,3,4 16 20 >> $O/overlap.txt 2>$O/overlap_err_${Q}_$H.txt
    echo "== GPU_MAX_HW_QUEUES=$Q BZK_MSM_HEAVY_PRIO=$H" >> $O/pipe.txt
    timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_${Q}_$H.txt
  done
done
cat $O/overlap.txt; cat $O/pipe.txt
export GPU_MAX_HW_QUEUES=16 BZK_MSM_HEAVY_PRIO=1
timeout 200 rocprofv3 --kernel-trace -d $O/trace_k2 -- python tools/overlap_probe.py 2 12 20 > $O/trace_k2.log 2>&1
T=$(find $O/trace_k2 -name "*.db" | head -1); python tools/trace_timeline.py $T auto 8 110 > $O/timeline_k2_heavy.txt 2>&1
sed -n 30,110p $O/timeline_k2_heavy.txt | cut -c1-120
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo finished
