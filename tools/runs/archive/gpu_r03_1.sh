set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_run1
timeout 1500 python -m pytest tests/test_gpu_mg.py tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r03_run1/pytest_mg_msm.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_run1/pytest_mg_msm.log
tail -5 gpurun_out/r03_run1/pytest_mg_msm.log
timeout 600 python bench.py --no-proofs --no-others > gpurun_out/r03_run1/bench_n1.log 2>&1; echo "rc=$?" >> gpurun_out/r03_run1/bench_n1.log
tail -c 3000 gpurun_out/r03_run1/bench_n1.log
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 --no-proofs > gpurun_out/r03_run1/bench_dry2.log 2>&1; echo "rc=$?" >> gpurun_out/r03_run1/bench_dry2.log
tail -c 2500 gpurun_out/r03_run1/bench_dry2.log
