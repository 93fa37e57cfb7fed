#!/bin/bash
# round-6 run 27: two-level fold of giant buckets for the G2 pair tails: parity, then stand-alone G2 MSM at 2^17 .. 2^20 points with and without it
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run27; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_mg.py -m gpu -q --timeout=420 --durations=4 -x -k "not 2p24 and not 2p26" ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -4 $O/pytest_msm.txt | cut -c1-200
timeout 100 python tests/tools/fuzz_gpu.py 30 2727 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
for rep in 1 2; do for nw in 1 0; do for lg in 17 18 19 20; do BZK_MSM_NO_WIDE_FOLD=$nw SWEEP_REPS=4 timeout 200 python tools/sweep.py child g2res $lg | grep '^{'; done; done; done > $O/g2_sizes.txt 2>&1
python - <<PY
import json
for l in open("$O/g2_sizes.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["log_n"], "no_wide", d.get("no_wide"), d["ms"], d["same_as_raw"], d["digest"], {k: v for k, v in d["prof"].items() if "fold" in k or "accum" in k})
PY
echo finished
