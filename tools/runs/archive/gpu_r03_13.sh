#!/bin/bash
# round-3 run 13: pipelined proofs/s with the producers' tree hashing on the host (default) vs on the device (BZK_BENCH_PRODUCER_DEV=1), alternating
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run13; mkdir -p $O
for V in 0 1 0 1; do
BZK_BENCH_PRODUCER_DEV=$V timeout 600 python bench.py --no-others --no-cpu-baseline --no-overlap > $O/bench_pd$V.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_pd$V.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("PRODUCER_DEV=$V", {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load","witness_s")})
PY
done
