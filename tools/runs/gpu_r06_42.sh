#!/bin/bash
# round-6 run 42: LAST state check (run 36's kernel sources - PMC stamps stay valid - with the thread names, the workers' dispatch setting and the final bench.py): whole GPU suite, smoke, default bench line, 4-rank rehearsal
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run42; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=420 --durations=6 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
( time timeout 900 python bench.py ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec","steps","warmup")}, d["roofline"]["avg_launch_ms"], d["roofline"]["traffic"], d["roofline"]["traffic_source"], d.get("two_msms_in_flight",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("two_processes",{}).get("proofs_per_s"), p.get("cpu_baseline"), p.get("process","")[:80])
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms"), o[k].get("roofline",{}).get("traffic")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p24","msm_g1_2p19","msm_g1_2p18","msm_g1_2p20_static_table") if k in o})
PY
tail -3 $O/bench_err.txt | cut -c1-300
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")})
print(json.dumps(d.get("proofs", {}).get("host_bound"))[:700])
PY
echo finished
