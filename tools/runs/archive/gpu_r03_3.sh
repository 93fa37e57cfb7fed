set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_state_compress.py -x -q -m gpu > $O/pytest_state.log 2>&1; echo "pytest rc=$?" >> $O/pytest_state.log
tail -15 $O/pytest_state.log
