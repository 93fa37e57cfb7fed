#!/bin/bash
# round-6 run 21: mid-size stand-alone G1 calls (2^16 .. 2^19): forced run lengths against the default task cut
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run21; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/sweep.py r6seg > $O/seg_sweep.txt 2>&1
python - <<PY
import json
for l in open("$O/seg_sweep.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["log_n"], "seg", d.get("seg"), d["mean_ms"], d["ms"], d["same_as_raw"], d["prof"])
    else:
        print(l.strip()[:200])
PY
echo finished
