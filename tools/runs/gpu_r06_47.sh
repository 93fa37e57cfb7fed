#!/bin/bash
# round-6 run 47: the default bench command and smoke on the final tree (HEAD after run 46)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run47; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
( time timeout 900 python bench.py ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["traffic_source"], d.get("cpu_baseline",{}).get("value"))
print({k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("process","")[:50])
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
PY
tail -2 $O/bench_err.txt | cut -c1-300
echo finished
