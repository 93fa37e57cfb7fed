#!/bin/bash
# round-5 run 10: the one-launch program allocated for three waves per SIMD (168 registers: a wave fits beside two waves of the G1 accumulation) instead of 512:
# parity, the kernel alone, the pipelined rate with staged deferred producers, same box as the plain reference legs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run10; mkdir -p $O
export TMPDIR=/tmp
( time timeout 240 python -m pytest tests/test_gpu_defer.py -m gpu -q -x ) > $O/pytest_defer.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_defer.txt; tail -6 $O/pytest_defer.txt | cut -c1-220
if [ $rc -ne 0 ]; then echo finished-early; exit 0; fi
for d in 0 1; do PROVE_DEFER=$d timeout 120 python tools/prove_serial.py 6 2>&1 | tail -1 | cut -c1-300; done > $O/prove_serial_ab.txt; cat $O/prove_serial_ab.txt
run() { echo "$*"; env "$@" timeout 200 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline 2>> $O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['proofs']
print({k: p.get(k) for k in ('witness_cpu_s', 'gpu_prove_s', 'proofs_per_s_pipelined', 'proofs_per_s_ring', 'prover_host_cpu_s_per_proof')}, {k: p['deferred'].get(k) for k in ('witness_cpu_s', 'gpu_prove_s')}, p.get('two_processes', {}).get('proofs_per_s'))
"; }
{ run BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1; run BZK_BENCH_DEFER=0 BZK_BENCH_STAGE=0; run BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1; run BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=0; } > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt | cut -c1-420
echo finished
