#!/bin/bash
# round-5 run 14: deferral for all three circuits on the device (Deposit / Withdraw through run_tx_bodies), the production block with plain / deferred legs for
# every work, then the DEFAULT bench command end to end on the final bench.py
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_defer.py -m gpu -q -x --timeout=200 --durations=4 ) > $O/pytest_defer.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_defer.txt; tail -10 $O/pytest_defer.txt | cut -c1-220
( time timeout 700 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"])
print({k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring")}, p.get("deferred"))
for k,v in pb.items():
    if isinstance(v,dict): print(k, {x:v.get(x) for x in ("decode_and_witness_s","prove_s","verified","deferred")})
    else: print(k, str(v)[:100])
print({k:(o[k].get("ms")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p24") if k in o})
PY
tail -4 $O/bench_err.txt | cut -c1-300
echo finished
