#!/bin/bash
# round-6 run 22: two-level fold of giant buckets: parity (MSM suites incl. the skewed-scalar cases), then A/B at 2^16 .. 2^21, unsplit and as two ranges
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run22; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_mg.py -m gpu -q --timeout=420 --durations=4 -x ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -8 $O/pytest_msm.txt | cut -c1-200
timeout 200 python tests/tools/fuzz_gpu.py 30 2222 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
timeout 1500 python tools/sweep.py r6wide > $O/wide_sweep.txt 2>&1
python - <<PY
import json
for l in open("$O/wide_sweep.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["log_n"], "no_wide", d.get("no_wide"), "split", d.get("split"), d["mean_ms"], d["ms"], d["same_as_raw"], {k: v for k, v in d["prof"].items() if "fold" in k or "accum" in k})
    else:
        print(l.strip()[:200])
PY
echo finished
