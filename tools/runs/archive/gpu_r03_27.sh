#!/bin/bash
# round-3 run 27: the default bench of the final build (driver-style), bounded
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run27; mkdir -p $O
timeout 420 python bench.py > $O/bench.txt 2> $O/bench_err.txt; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("default", d["value"], d["ms_per_step"], d["roofline"]["traffic"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load")}, p["host_waits"]["mode"])
print(sorted(d["other_configs"].keys()))
PY
