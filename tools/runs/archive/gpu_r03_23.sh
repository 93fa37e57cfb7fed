#!/bin/bash
# round-3 run 23: the pipelined proofs/s of the bench under the box's 16-CPU quota: blocking waits, fewer producer threads
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run23; mkdir -p $O
run() {
  L=$1; shift
  env "$@" timeout 400 python bench.py --no-others --no-overlap --no-cpu-baseline > $O/bench_$L.txt 2>&1
  python - <<PY
import json
d=json.loads(open("$O/bench_$L.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("$L", {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load")})
PY
}
for rep in 1 2; do
run default BZK_X=0
run blocking BZK_SYNC_BLOCKING=1
run prod4 BZK_BENCH_PROD_THREADS=4
run prod4_blocking BZK_BENCH_PROD_THREADS=4 BZK_SYNC_BLOCKING=1
run p6t8_blocking BZK_BENCH_PRODUCERS=6 BZK_SYNC_BLOCKING=1
done 2>&1 | grep -v "^+" | tee $O/ab.txt
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
