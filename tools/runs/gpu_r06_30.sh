#!/bin/bash
# round-6 run 30: row / column sums with 2 / 4 / 16 buckets per lane in the serial phase against 8
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run30; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/sweep.py r6rcleaf > $O/rcleaf.txt 2>&1
python - <<PY
import json
for l in open("$O/rcleaf.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["log_n"], "leaf", d.get("rc_leaf"), d["mean_ms"], d["ms"], d["same_as_raw"], {k: v for k, v in d["prof"].items() if "rowcol" in k or "bitsum" in k})
    else:
        print(l.strip()[:200])
PY
echo finished
