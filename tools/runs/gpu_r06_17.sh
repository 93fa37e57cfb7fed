#!/bin/bash
# round-6 run 17: the driver's default bench command on the final bench.py (producers stage their instances by default), and the four-rank rehearsal
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run17; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["traffic_source"], d.get("two_msms_in_flight",{}).get("value"))
print({k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, {k:v for k,v in p.get("deferred",{}).items() if k.startswith("live")}, p.get("two_processes",{}).get("proofs_per_s"))
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
PY
tail -3 $O/bench_err.txt | cut -c1-300
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")}, {k:v for k,v in d["proofs"].get("deferred",{}).items() if k.startswith("live")})
PY
echo finished
