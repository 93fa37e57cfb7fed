#!/bin/bash
# round-3 run 31: bench with the informational two-process proof measurement (a second prover process on the same GPU), bounded
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run31; mkdir -p $O
timeout 200 python bench.py --no-others --no-overlap --no-cpu-baseline > $O/bench_quick.txt 2> $O/bench_quick_err.txt; echo "rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_quick.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("quick", d["value"], d["ms_per_step"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","two_processes")})
PY
tail -3 $O/bench_quick_err.txt | cut -c1-300
