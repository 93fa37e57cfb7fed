#!/bin/bash
# round-5 run 15: the N > 1 launch path on the final bench.py, rehearsed with four ranks on ONE GPU (gloo for the CPU-side votes, the device group's shared-memory
# transport): the line the driver's SCALE tier would parse, with every rank's bzk_mg_stats
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run15; mkdir -p $O
export TMPDIR=/tmp
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")})
print(json.dumps(d.get("collective"))[:1500])
print(json.dumps(d.get("proofs"))[:900])
PY
tail -5 $O/bench_dryrun_gpus4_err.txt | cut -c1-300
echo finished
