#!/bin/bash
# round-4 run 2: new mg / devtree tests (error propagation, rollback), default bench with production_block + ring + quota-sized
# cpu_baseline, 4-rank gloo rehearsal of the N>1 line with the proofs leg, one-rank-of-8 emulation, serial per-kernel table of a proof
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mg.py tests/test_gpu_mpn_devtree.py -m gpu -q -x > $O/pytest_mg_devtree.txt 2>&1; echo "rc=$?" >> $O/pytest_mg_devtree.txt
( time timeout 1200 python bench.py ) > $O/bench.txt 2> $O/bench_err.txt
timeout 600 python tools/sweep.py r4rank8 > $O/rank_of_8.txt 2>&1
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 900 python bench.py --gpus 4 --steps 10 --warmup 2 > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
BZK_PROVE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial_trace.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
tail -5 $O/pytest_mg_devtree.txt; tail -5 $O/bench_err.txt; python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["cpu_baseline"])
print(json.dumps(d["other_configs"].get("production_block"))[:3000])
p=d["proofs"]; print({k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring","witness_s")}, p.get("cpu_baseline"), p.get("two_processes"))
PY
cat $O/rank_of_8.txt | cut -c1-600; tail -3 $O/bench_dryrun_gpus4_err.txt; cut -c1-3000 $O/bench_dryrun_gpus4.txt; tail -2 $O/serial_trace.log; head -40 $O/serial_trace_summary.txt
echo finished
