#!/bin/bash
# round-5 run 20: the default bench command on the FINAL bench.py (adds proofs.deferred.with_device_builder: deferred generator + the transition builder's tree walk on the device)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run20; mkdir -p $O
export TMPDIR=/tmp
( time timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"])
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring")})
print(json.dumps(p.get("deferred"))[:700])
print({k:(v.get("prove_s"), v.get("verified")) for k,v in d["other_configs"]["production_block"].items() if isinstance(v,dict)})
PY
tail -4 $O/bench_err.txt | cut -c1-300
echo finished
