#!/bin/bash
# round-3 run 29: N = 2 dry run of the launch path on the last build (python bench.py --gpus 2, self-spawned; ranks share the GPU), bounded
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run29; mkdir -p $O
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 240 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_dryrun_gpus2.txt 2> $O/bench_dryrun_gpus2_err.txt; echo "rc=$?"
tail -1 $O/bench_dryrun_gpus2.txt | cut -c1-400
python - <<PY
import json
d=json.loads(open("$O/bench_dryrun_gpus2.txt").read().strip().splitlines()[-1])
print(d["value"], d["n_gpus"], d.get("collective"), d.get("proofs_per_sec"), d["proofs"].get("per_rank_pipelined"), d["proofs"].get("host_waits"))
PY
