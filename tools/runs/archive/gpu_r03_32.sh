#!/bin/bash
# round-3 run 32 / 36: the default bench of the last commit, exactly as the driver runs it (bounded); O overridable
set -x
cd $GRAFT_REPO_ROOT
O=${O:-gpurun_out/r03_run32}; mkdir -p $O
timeout 420 python bench.py > $O/bench.txt 2> $O/bench_err.txt; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("default", d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["frac"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined")}, p.get("two_processes",{}).get("proofs_per_s"), d["cpu_baseline"]["value"], sorted(d["other_configs"].keys())[:3])
PY
