set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run2; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
tail -c 6000 $O/bench_default.log
