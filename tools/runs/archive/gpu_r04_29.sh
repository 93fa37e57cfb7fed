#!/bin/bash
# round-4 run 29: last state check of the round (host IFMA witness path, device state, native worker, any-width cooperative Poseidon):
# full GPU suite, smoke, fuzz, kernel trace of the headline command, default bench, one-slot worker rate
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run29; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 300 python tests/tools/fuzz_gpu.py 60 123 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
timeout 300 python tools/worker_rate.py 24 > $O/worker_rate.txt 2>&1
tail -10 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; tail -2 $O/fuzz.txt | cut -c1-500; head -8 $O/trace_summary.txt | cut -c1-150; tail -3 $O/worker_rate.txt | cut -c1-400
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o["production_block"]
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"]["traffic_source"], {k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")}, p["two_processes"]["proofs_per_s"])
print({k:(v.get("prove_s"),v.get("decode_and_witness_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms"), o[k].get("roofline",{}).get("traffic")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","state_device_incremental")})
PY
echo finished
