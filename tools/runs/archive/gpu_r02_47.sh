#!/bin/bash
# round-2 run 47: last build (BZK_SYNC_BLOCKING env in bzk_ctx_create, cpu_quota in the bench line): smoke, a test subset, default bench
O=gpurun_out/r02_47
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 300 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_worker.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 400 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -2 $O/smoke.txt; tail -3 $O/pytest.txt; python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print(d["value"], d["ms_per_step"], d["proofs_per_sec"], d["cpu_baseline"], p["cpu_baseline"], d["roofline"]["traffic"])
PY
echo finished
