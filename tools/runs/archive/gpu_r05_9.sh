#!/bin/bash
# round-5 run 9: staging - uploads + the deferred-value program on the PRODUCER's context (bzk_r1cs_stage), the prover slots copy device to device
# (bzk_groth16_prove_staged): parity, then the pipelined rate with live producers: plain / staged plain / staged deferred, alternating, same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run9; mkdir -p $O
export TMPDIR=/tmp
( time timeout 240 python -m pytest tests/test_gpu_defer.py -m gpu -q -x --durations=3 ) > $O/pytest_defer.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_defer.txt; tail -12 $O/pytest_defer.txt | cut -c1-220
if [ $rc -ne 0 ]; then echo finished-early; exit 0; fi
run() { echo "$*"; env "$@" timeout 200 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline 2>> $O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['proofs']
print({k: p.get(k) for k in ('witness_cpu_s', 'gpu_prove_s', 'proofs_per_s_pipelined', 'proofs_per_s_ring', 'prover_host_cpu_s_per_proof')}, {k: p['deferred'].get(k) for k in ('witness_cpu_s', 'gpu_prove_s')}, p.get('two_processes', {}).get('proofs_per_s'))
"; }
{ for rep in 1 2; do run BZK_BENCH_DEFER=0 BZK_BENCH_STAGE=0; run BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1; run BZK_BENCH_DEFER=0 BZK_BENCH_STAGE=1; done; run BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1 BZK_BENCH_SLOTS=3; } > $O/bench_stage_ab.txt 2>&1
cat $O/bench_stage_ab.txt | cut -c1-420; tail -3 $O/bench_err.txt | cut -c1-300
echo finished
