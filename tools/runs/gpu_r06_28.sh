#!/bin/bash
# round-6 run 28: stand-alone MSM time against size, 2^12 .. 2^24 (G1) and 2^12 .. 2^21 (G2): are there other size cliffs? + the new giant-bucket tests
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run28; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q --timeout=420 -x -k "giant or split" ) > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt; tail -3 $O/pytest_new.txt
for lg in 12 13 14 15 16 17 18 19 20 21 22 23 24; do SWEEP_REPS=5 timeout 200 python tools/sweep.py child g1res $lg | grep '^{'; done > $O/g1_sizes.txt 2>&1
for lg in 12 13 14 15 16 17 18 19 20 21; do SWEEP_REPS=3 timeout 200 python tools/sweep.py child g2res $lg | grep '^{'; done > $O/g2_sizes.txt 2>&1
python - <<PY
import json
for f in ("g1_sizes", "g2_sizes"):
    for l in open("$O/%s.txt" % f):
        if l.startswith("{"):
            d = json.loads(l); print(d["mode"], d["log_n"], d["ms"], d["Mpt/s"], d["same_as_raw"], d["prof"])
PY
echo finished
