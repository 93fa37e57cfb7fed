#!/bin/bash
# round-3 run 24: the default bench of the final build (driver-style), twice
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run24; mkdir -p $O
for i in 1 2; do
timeout 900 python bench.py > $O/bench_$i.txt 2> $O/bench_err_$i.txt
python - <<PY
import json
d=json.loads(open("$O/bench_$i.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("run $i", d["value"], d["ms_per_step"], d["roofline"]["traffic"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load","host_waits")})
PY
done
