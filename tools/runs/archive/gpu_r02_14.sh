#!/bin/bash
# round-2 run 14: prover slots x reduce chunk under pipelined load: does less reduce work (ch = 16) pay once more slots hide its longer chain?
# plus the new worker test with bellman parameter files
set -x
O=gpurun_out/r02_14
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_worker.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
for s in 4 6 8; do for ch in 8 16; do
  BZK_BENCH_SLOTS=$s BZK_MSM_CHUNK=$ch timeout 300 python bench.py --steps 3 --warmup 1 --no-others --no-cpu-baseline --no-overlap > $O/b_${s}_${ch}.txt 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_${s}_${ch}.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("slots $s chunk $ch: pipelined", p.get("proofs_per_s_pipelined"), "gpu_prove_s", p.get("gpu_prove_s"))
PY
done; done 2>&1 | grep "slots" | tee $O/summary.txt
tail -3 $O/pytest.txt
echo finished
