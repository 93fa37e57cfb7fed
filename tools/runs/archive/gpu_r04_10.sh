#!/bin/bash
# round-4 run 10: is the pipelined-proofs leg GPU-bound?  kernel trace of the proofs section, union of kernel intervals in the busiest second
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run10; mkdir -p $O
export TMPDIR=/tmp
BZK_BENCH_TWO_PROCS=0 timeout 600 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline > $O/bench_traced.txt 2> $O/bench_traced_err.txt
T=$(find $O/trace -name "*.db" | head -1)
python tools/trace_busy.py $T 1.0 > $O/busy.txt 2>&1
python tools/trace_busy.py $T 0.25 >> $O/busy.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cat $O/busy.txt; python - <<PY
import json
d=json.loads(open("$O/bench_traced.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print({k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring")})
PY
echo finished
