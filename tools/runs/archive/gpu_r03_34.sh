#!/bin/bash
# round-3 run 34: last build after the host witness fast path: full GPU suite, smoke, then the bench's proof section twice (bounded)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run34; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
for i in 1 2; do
timeout 150 python bench.py --no-others --no-overlap --no-cpu-baseline > $O/bench_quick_$i.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_quick_$i.txt").read().strip().splitlines()[-1])
p=d["proofs"]; print("quick $i", d["value"], {k:p.get(k) for k in ("witness_s","gpu_prove_s","proofs_per_s_pipelined","producer_synth_s_mean_under_load")}, p.get("two_processes",{}).get("proofs_per_s"))
PY
done
