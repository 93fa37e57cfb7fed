#!/bin/bash
# round-4 run 13: last state check of the committed build: full GPU suite, smoke, default bench (stamps of both PMC files must match)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run13; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
tail -8 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; tail -4 $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); o=d["other_configs"]; p=d["proofs"]
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print({k:(o[k].get("ms"), o[k].get("roofline",{}).get("traffic")) for k in ("tree_2p24","ntt_2p20","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p20_static_table","msm_g1_2p24")})
print({k:p.get(k) for k in ("gpu_prove_s","witness_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")}, p.get("two_processes",{}).get("proofs_per_s"), d["cpu_baseline"]["value"], p["cpu_baseline"]["value"])
pb=o["production_block"]; print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
PY
echo finished
