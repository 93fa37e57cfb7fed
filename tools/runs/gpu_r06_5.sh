#!/bin/bash
# round-6 run 5: the deferred-witness program on cooperating lanes (BZK_WF_MODE=coop, the new default): (A) parity (fixtures of the independent restatement, the
# oracle prover, the workers); (B) what it costs a proof (kernel trace of tools/prove_serial.py with PROVE_DEFER=1, both forms); (C) the bench's proofs section with
# plain / deferred + staged live producers, both forms of the program
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run5; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_defer.py tests/test_gpu_worker.py -m gpu -q --timeout=420 --durations=5 ) > $O/pytest_defer.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_defer.txt
tail -12 $O/pytest_defer.txt | cut -c1-200
for M in coop one; do
  BZK_WF_MODE=$M PROVE_DEFER=1 BZK_PROVE_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$M -- python tools/prove_serial.py 6 > $O/trace_$M.log 2>&1
  T=$(find $O/trace_$M -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_${M}_summary.txt 2>&1
  grep -E "wf_|kernel  " $O/trace_${M}_summary.txt | cut -c1-150
  tail -3 $O/trace_$M.log | cut -c1-300
done
B="python bench.py --steps 10 --warmup 3 --no-others --no-cpu-baseline --no-overlap"
( time timeout 600 $B ) > $O/bench_plain.txt 2> $O/bench_plain_err.txt
( time BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1 timeout 600 $B ) > $O/bench_defer_coop.txt 2> $O/bench_defer_coop_err.txt
( time BZK_WF_MODE=one BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1 timeout 600 $B ) > $O/bench_defer_one.txt 2> $O/bench_defer_one_err.txt
( time BZK_BENCH_DEFER=1 timeout 600 $B ) > $O/bench_defer_coop_nostage.txt 2> $O/bench_defer_coop_nostage_err.txt
python - <<PY
import json
for n in ("plain","defer_coop","defer_one","defer_coop_nostage"):
    try:
        d=json.loads(open("$O/bench_%s.txt"%n).read().strip().splitlines()[-1]); p=d["proofs"]
        print(n, {k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("deferred"))
    except Exception as e:
        print(n, "failed", e)
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo finished
