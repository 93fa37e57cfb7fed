#!/bin/bash
# round-2 run 50: default bench with the 2^24 G1 MSM line added to other_configs
O=gpurun_out/r02_50
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T0=$(date +%s); timeout 400 python bench.py > $O/bench.txt 2> $O/bench_err.txt; echo "bench wall $(( $(date +%s) - T0 )) s"

python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["proofs_per_sec"]); print(d["other_configs"].get("msm_g1_2p24")); print({k: d["other_configs"][k].get("ms") for k in d["other_configs"]})
PY
echo finished
