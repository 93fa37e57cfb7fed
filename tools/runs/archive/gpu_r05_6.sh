#!/bin/bash
# round-5 run 6: the deferred witness values on the device (witfill.hip) - parity first (same proof bytes as the plain path), then what it buys: the bench's
# proofs section with live producers on the plain / on the deferred generator, same box, alternating; the G2 MSM with the two-level reduction for every call
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run6; mkdir -p $O
export TMPDIR=/tmp
export SWEEP_CHILD_TIMEOUT=60
( time timeout 240 python -m pytest tests/test_gpu_defer.py -m gpu -q -x --durations=5 ) > $O/pytest_defer.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_defer.txt; tail -25 $O/pytest_defer.txt | cut -c1-220
if [ $rc -ne 0 ]; then BZK_DEBUG=1 timeout 120 python -m pytest tests/test_gpu_defer.py -m gpu -q -x -k "update_3_3_1" > $O/debug_launches.txt 2>&1; tail -40 $O/debug_launches.txt | cut -c1-200; fi
for rep in 1 2; do for d in 0 1; do echo "BZK_BENCH_DEFER=$d"; BZK_BENCH_DEFER=$d timeout 200 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline 2> $O/bench_err_$d.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['proofs']
print({k: p.get(k) for k in ('witness_s', 'witness_cpu_s', 'gpu_prove_s', 'proofs_per_s_serial', 'proofs_per_s_pipelined', 'proofs_per_s_ring', 'prover_host_cpu_s_per_proof')}, p.get('deferred'), p.get('two_processes', {}).get('proofs_per_s'))
"; done; done > $O/bench_defer_ab.txt 2>&1
cat $O/bench_defer_ab.txt | cut -c1-900; tail -3 $O/bench_err_1.txt | cut -c1-300
timeout 200 python tools/sweep.py r5g2b > $O/g2_two_level.txt 2>&1
cut -c1-500 $O/g2_two_level.txt
echo finished
