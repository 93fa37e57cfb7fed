#!/bin/bash
# round-3 run 17: (a) host field on 64-bit limbs for the Horner / to-affine / proof assembly, (b) quad level-2 bucket reduction:
# parity, A/B of the MSM headline over the reduction forms, then the default bench (proofs now with full-size r, s)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_mg.py tests/test_gpu_groth16.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
BZK_MSM_REDUCE2=1 timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu > $O/pytest_reduce2.txt 2>&1; echo "rc=$?" >> $O/pytest_reduce2.txt; tail -3 $O/pytest_reduce2.txt
run() {  # label, env...
  L=$1; shift
  env "$@" timeout 300 python bench.py --no-proofs --no-others --no-overlap --no-cpu-baseline > $O/bench_$L.txt 2>&1
  python - <<PY
import json
d=json.loads(open("$O/bench_$L.txt").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; print("$L", d["value"], d["ms_per_step"], {x:k.get(x) for x in ("msm_accumulate","msm_reduce","msm_reduce_l2","msm_window_partial","msm_window_sum")})
PY
}
for rep in 1 2; do
run one_level BZK_MSM_REDUCE2=-1
run two_level_lane BZK_MSM_REDUCE2=1 BZK_MSM_QUAD_L2=0
run two_level_quad4 BZK_MSM_REDUCE2=1 BZK_MSM_L2_CH=4
run two_level_quad2 BZK_MSM_REDUCE2=1 BZK_MSM_L2_CH=2
run two_level_quad8 BZK_MSM_REDUCE2=1 BZK_MSM_L2_CH=8
done 2>&1 | grep -v "^+" | tee $O/ab.txt
timeout 900 python bench.py > $O/bench_default.txt 2>$O/bench_default.err; tail -c 3000 $O/bench_default.txt
