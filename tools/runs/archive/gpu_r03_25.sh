#!/bin/bash
# round-3 run 25: bench pipelined-proof section with blocking waits switched on mid-process (env read at slot-context creation)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run25; mkdir -p $O
timeout 200 python bench.py --no-others --no-overlap --no-cpu-baseline > $O/bench_quick.txt 2> $O/bench_quick_err.txt; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_quick.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("quick", d["value"], d["ms_per_step"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load","host_waits","pipeline")})
PY
