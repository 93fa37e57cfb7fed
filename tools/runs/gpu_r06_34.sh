#!/bin/bash
# round-6 run 34: one rank of 8 (2 of 16 windows over 2^23 points, resident set): run length per task forced against the default; SHARDS=4 and 2 as well
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run34; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for sh in 8 4 2; do
for seg in "" 24 32 48 64 96 128; do
  lg=$((20 + $(python -c "import math;print(int(math.log2($sh)))")))
  BZK_MSM_SEG=$seg SHARDS=$sh timeout 200 python tools/sweep.py child g1winres $lg | grep '^{' | sed "s/^{/{\"seg\": \"$seg\", \"shards\": $sh, /"
done; done; done > $O/rank_seg.txt 2>&1
python - <<PY
import json
for l in open("$O/rank_seg.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["shards"], d["log_n"], "seg", d["seg"], d["ms"], d["prof"])
PY
echo finished
