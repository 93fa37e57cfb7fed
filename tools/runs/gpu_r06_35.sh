#!/bin/bash
# round-6 run 35: the task cut left off for two tasks per lane of full buckets: ranks of 2 / 4 / 8 and the stand-alone sizes again + window / mg parity
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run35; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_mg.py -m gpu -q --timeout=420 -x -k "not 2p26" ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt | cut -c1-200
for rep in 1 2; do
for sh in 8 4 2; do lg=$((20 + $(python -c "import math;print(int(math.log2($sh)))"))); SHARDS=$sh timeout 200 python tools/sweep.py child g1winres $lg | grep '^{' | sed "s/^{/{\"shards\": $sh, /"; done
for lg in 19 20 21 22; do SWEEP_REPS=6 timeout 200 python tools/sweep.py child g1res $lg | grep '^{' | sed "s/^{/{\"shards\": 1, /"; done
done > $O/ranks.txt 2>&1
python - <<PY
import json
for l in open("$O/ranks.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["shards"], d["log_n"], d["ms"], {k: v for k, v in d["prof"].items() if "acc" in k or "fold" in k})
PY
echo finished
