#!/bin/bash
# round-6 run 6: the cooperative witness fill with all dense traces in ONE launch: parity, its kernel times, and the bench's proofs section twice each with plain and
# with deferred + staged live producers (alternating: the pipelined figures of one box scatter by ~ +-1.5 %)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run6; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_defer.py -m gpu -q --timeout=420 ) > $O/pytest_defer.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_defer.txt
tail -5 $O/pytest_defer.txt | cut -c1-200
PROVE_DEFER=1 BZK_PROVE_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_coop -- python tools/prove_serial.py 6 > $O/trace_coop.log 2>&1
T=$(find $O/trace_coop -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_coop_summary.txt 2>&1
grep -E "wf_|calls" $O/trace_coop_summary.txt | cut -c1-150
B="python bench.py --steps 10 --warmup 3 --no-others --no-cpu-baseline --no-overlap"
for rep in 1 2; do
  ( time timeout 600 $B ) > $O/bench_plain_$rep.txt 2> $O/bench_plain_err_$rep.txt
  ( time BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1 timeout 600 $B ) > $O/bench_defer_$rep.txt 2> $O/bench_defer_err_$rep.txt
done
( time BZK_BENCH_DEFER=1 BZK_BENCH_STAGE=1 BZK_BENCH_PRODUCER_DEV=1 timeout 600 $B ) > $O/bench_defer_dev_1.txt 2> $O/bench_defer_dev_err_1.txt
python - <<PY
import json
for n in ("plain_1","defer_1","plain_2","defer_2","defer_dev_1"):
    try:
        d=json.loads(open("$O/bench_%s.txt"%n).read().strip().splitlines()[-1]); p=d["proofs"]
        print(n, d["value"], {k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, {k:v for k,v in p.get("deferred",{}).items() if k in ("witness_cpu_s","gpu_prove_s")})
    except Exception as e:
        print(n, "failed", e)
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo finished
