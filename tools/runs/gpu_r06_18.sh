#!/bin/bash
# round-6 run 18: one stand-alone G1 MSM call as several window ranges in flight (msm_run_split): parity, then the sweep (ranges x child priority), then the headline command
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run18; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -m gpu -q --timeout=420 --durations=4 -x ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -8 $O/pytest_msm.txt | cut -c1-200
timeout 1200 python tools/sweep.py r6split > $O/split_sweep.txt 2>&1
cut -c1-330 $O/split_sweep.txt
for sp in 1 2; do
( time BZK_MSM_SPLIT=$sp timeout 600 python bench.py --steps 20 --warmup 5 --no-proofs --no-others --no-cpu-baseline ) > $O/bench_headline_split$sp.txt 2> $O/bench_headline_err_split$sp.txt
python - <<PY
import json
d=json.loads(open("$O/bench_headline_split$sp.txt").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_launch_ms"], d.get("kernel_ms_per_step"), d.get("two_msms_in_flight"))
PY
done
echo finished
