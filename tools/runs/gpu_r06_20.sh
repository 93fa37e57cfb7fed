#!/bin/bash
# round-6 run 20: split ranges, priority modes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run20; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/sweep.py r6split3 > $O/split_sweep.txt 2>&1
python - <<PY
import json
for l in open("$O/split_sweep.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["log_n"], d.get("split"), d.get("cuts"), d.get("split_prio"), d["mean_ms"], d["ms"], d["same_as_raw"], {k: v for k, v in d["prof"].items() if v > 0.25})
    else:
        print(l.strip()[:200])
PY
echo finished
