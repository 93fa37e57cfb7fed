#!/bin/bash
# round-6 run 16: live producers STAGING their (plain) instances on a context of their own (BZK_BENCH_STAGE=1 without deferral) against the default, alternating
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run16; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-others --no-cpu-baseline --no-overlap"
for rep in 1 2; do
  ( timeout 600 $B ) > $O/bench_plain_$rep.txt 2> $O/bench_plain_err_$rep.txt
  ( BZK_BENCH_STAGE=1 timeout 600 $B ) > $O/bench_stage_$rep.txt 2> $O/bench_stage_err_$rep.txt
done
python - <<PY
import json
for n in ("plain_1","stage_1","plain_2","stage_2"):
    try:
        d=json.loads(open("$O/bench_%s.txt"%n).read().strip().splitlines()[-1]); p=d["proofs"]
        print(n, d["value"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("deferred",{}).get("live_producers_stage"))
    except Exception as e:
        print(n, "failed", e)
PY
echo finished
