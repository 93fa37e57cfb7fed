#!/bin/bash
# round-3 run 16: quad-cooperative window-sum trees (G1): parity, then A/B of the MSM headline (BZK_MSM_QUAD_TREE=0|1, alternating)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_mg.py tests/test_gpu_groth16.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for V in 0 1 0 1; do
BZK_MSM_QUAD_TREE=$V timeout 300 python bench.py --no-proofs --no-others --no-overlap --no-cpu-baseline > $O/bench_q$V.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_q$V.txt").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; print("QUAD_TREE=$V", d["value"], d["ms_per_step"], {x:k[x] for x in ("msm_accumulate","msm_reduce","msm_window_partial","msm_window_sum")})
PY
done
