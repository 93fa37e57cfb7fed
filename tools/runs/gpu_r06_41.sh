#!/bin/bash
# round-6 run 41: as run 40 with the proofs child run BEFORE the parent touches the GPU
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run41; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_worker.py -m gpu -q --timeout=420 -x ) > $O/pytest_worker.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_worker.txt; tail -3 $O/pytest_worker.txt | cut -c1-200
for rep in 1 2; do
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_$rep.txt 2> $O/bench_err_$rep.txt
python - <<PY
import json
d=json.loads(open("$O/bench_$rep.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"])
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof","prover_host_cpu_cores_busy")}, p.get("two_processes",{}).get("proofs_per_s"), (p.get("cpu_baseline") or {}).get("value"), p.get("process","")[:60], p.get("deferred",{}).get("gpu_prove_s"))
PY
tail -2 $O/bench_err_$rep.txt | cut -c1-300
done
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")}, d["proofs"].get("process","")[:60])
print(json.dumps(d.get("proofs", {}).get("host_bound"))[:900])
PY
tail -3 $O/bench_dryrun_gpus4_err.txt | cut -c1-300
echo finished
