#!/bin/bash
# round-5 run 8: the deferred-witness program as ONE launch per proof (wf_tx_kernel: a workgroup of four waves per transition, barriers between the
# dependency levels): parity, one proof at a time, then the pipelined rate with live producers on the plain / deferred generator, same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run8; mkdir -p $O
export TMPDIR=/tmp
( time timeout 200 python -m pytest tests/test_gpu_defer.py -m gpu -q -x --durations=3 ) > $O/pytest_defer.txt 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_defer.txt; tail -9 $O/pytest_defer.txt | cut -c1-220
if [ $rc -ne 0 ]; then BZK_WF_MODE=levels timeout 100 python -m pytest tests/test_gpu_defer.py -m gpu -q -x 2>&1 | tail -3; echo finished-early; exit 0; fi
for d in 0 1; do PROVE_DEFER=$d timeout 120 python tools/prove_serial.py 6 2>&1 | tail -1 | cut -c1-300; done > $O/prove_serial_ab.txt; cat $O/prove_serial_ab.txt
PROVE_DEFER=1 timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/prove_serial.py 6 > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/deferred_proofs_kernel_table.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
grep -E "wf_|calls" $O/deferred_proofs_kernel_table.txt | cut -c1-150
run() { echo "$*"; env "$@" timeout 200 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline 2>> $O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['proofs']
print({k: p.get(k) for k in ('witness_cpu_s', 'gpu_prove_s', 'proofs_per_s_pipelined', 'proofs_per_s_ring')}, {k: p['deferred'].get(k) for k in ('witness_cpu_s', 'gpu_prove_s')}, p.get('two_processes', {}).get('proofs_per_s'))
"; }
{ run BZK_BENCH_DEFER=0; run BZK_BENCH_DEFER=1; run BZK_BENCH_DEFER=1 BZK_BENCH_SLOTS=6; run BZK_BENCH_DEFER=1 BZK_BENCH_SLOTS=8; run BZK_BENCH_DEFER=0; run BZK_BENCH_DEFER=1 BZK_BENCH_SLOTS=6 BZK_WF_PRIO=1; } > $O/bench_defer_ab.txt 2>&1
cat $O/bench_defer_ab.txt | cut -c1-400; tail -3 $O/bench_err.txt | cut -c1-300
echo finished
