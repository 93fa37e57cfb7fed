#!/bin/bash
# round-3 run 30: prover slots per process on the last build (the N = 2 dry run showed 65 proofs/s from two 4-slot processes on one GPU)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run30; mkdir -p $O
run() {
  L=$1; shift
  env "$@" timeout 150 python bench.py --no-others --no-overlap --no-cpu-baseline > $O/bench_$L.txt 2>&1
  python - <<PY
import json
d=json.loads(open("$O/bench_$L.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("$L", {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load")})
PY
}
for rep in 1 2; do
run slots4 BZK_BENCH_SLOTS=4
run slots6 BZK_BENCH_SLOTS=6
run slots8 BZK_BENCH_SLOTS=8
run slots8_p10 BZK_BENCH_SLOTS=8 BZK_BENCH_PRODUCERS=10
done 2>&1 | grep -v "^+" | tee $O/ab.txt
