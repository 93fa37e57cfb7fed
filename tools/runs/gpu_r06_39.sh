#!/bin/bash
# round-6 run 39: AMD_DIRECT_DISPATCH=0 (launches handed to the runtime's per-stream worker threads; run 38: prover host CPU per proof 0.0174 -> 0.0065): what it does to the
# stand-alone headline, to the pipelined proofs with live producers and to the small-call latency
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run39; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for dd in 1 0; do
( AMD_DIRECT_DISPATCH=$dd timeout 600 python bench.py --steps 20 --warmup 5 --no-proofs --no-others --no-cpu-baseline ) > $O/headline_dd${dd}_$rep.txt 2> $O/headline_dd${dd}_${rep}_err.txt
python - <<PY
import json
d=json.loads(open("$O/headline_dd${dd}_$rep.txt").read().strip().splitlines()[-1])
print("direct_dispatch=$dd", {k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_launch_ms"], d.get("two_msms_in_flight",{}).get("in_flight_2"), d.get("two_msms_in_flight",{}).get("in_flight_4"))
PY
done; done
for dd in 1 0; do for lg in 12 16 18; do AMD_DIRECT_DISPATCH=$dd SWEEP_REPS=8 SWEEP_MEAN_OVER=20 timeout 200 python tools/sweep.py child g1res $lg | grep '^{' | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('direct_dispatch=$dd small call', d['log_n'], d['mean_ms'], d['ms'])"; done; done
for dd in 1 0; do
( AMD_DIRECT_DISPATCH=$dd timeout 900 python bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline ) > $O/bench_dd$dd.txt 2> $O/bench_dd${dd}_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench_dd$dd.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("direct_dispatch=$dd", {k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, {k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof","prover_host_cpu_cores_busy")}, p.get("two_processes",{}).get("proofs_per_s"))
PY
done
echo finished
