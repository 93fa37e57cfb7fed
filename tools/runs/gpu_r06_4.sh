#!/bin/bash
# round-6 run 4: the multiplication-free bucket reduction of the G1 MSM (msm_impl.cuh section 6b): (A) parity - every MSM / endomorphism / proof test against the
# oracle; (B) same-box A/B against the chunked running sum (BZK_MSM_BITSUM=0): stand-alone MSMs of several sizes, MSMs in flight, the pipelined proofs ceiling
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run4; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_fullsize.py tests/test_gpu_mpn_prove.py tests/test_gpu_groth16.py -m gpu -q --timeout=420 --durations=6 -x ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -14 $O/pytest_msm.txt | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 900 python tools/sweep.py r6bitsum > $O/bitsum_ab.txt 2>&1
cut -c1-400 $O/bitsum_ab.txt
for B in 0 1; do
  export BZK_MSM_BITSUM=$B
  echo "== BZK_MSM_BITSUM=$B" >> $O/overlap.txt
  timeout 200 python tools/overlap_probe.py 1,2,4 16 20 >> $O/overlap.txt 2>$O/overlap_err_$B.txt
  echo "== BZK_MSM_BITSUM=$B" >> $O/pipe.txt
  timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_$B.txt
done
cat $O/overlap.txt $O/pipe.txt
unset BZK_MSM_BITSUM
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-proofs --no-others --no-cpu-baseline ) > $O/bench_headline.txt 2> $O/bench_headline_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench_headline.txt").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_launch_ms"], d.get("kernel_ms_per_step"), d.get("two_msms_in_flight"))
PY
tail -3 $O/bench_headline_err.txt | cut -c1-300
echo finished
