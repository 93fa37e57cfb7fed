#!/bin/bash
# round-6 run 14: the device groups exchange the TERMS of their windows' bucket sets (G1): parity of every group form on one GPU (contexts, processes, RCCL one-rank, the
# injected fault), what one rank of 8 costs now (2 windows over 2^23 points), the four-rank rehearsal of the bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_mg.py tests/test_gpu_msm.py -m gpu -q --timeout=420 --durations=5 -k "not 2p26" ) > $O/pytest_mg.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_mg.txt
tail -10 $O/pytest_mg.txt | cut -c1-200
for B in 0 1; do
  echo "== BZK_MSM_BITSUM=$B: one rank of 8 (2 of 16 windows over 2^23 points, resident set): tools/sweep.py child g1winres 23" >> $O/rank_of_8.txt
  BZK_MSM_BITSUM=$B SHARDS=8 timeout 300 python tools/sweep.py child g1winres 23 >> $O/rank_of_8.txt 2>&1
done
cut -c1-500 $O/rank_of_8.txt
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")})
print(json.dumps(d.get("collective"))[:1800])
PY
tail -3 $O/bench_dryrun_gpus4_err.txt | cut -c1-300
echo finished
