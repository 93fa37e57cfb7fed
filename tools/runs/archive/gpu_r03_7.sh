set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run7; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mpn_devtree.py -x -q -m gpu > $O/pytest_devtree.log 2>&1; echo "rc=$?" >> $O/pytest_devtree.log; tail -15 $O/pytest_devtree.log
timeout 900 python tools/witness_ab.py 256 > $O/witness_ab_256.log 2>&1; cat $O/witness_ab_256.log | tail -3
timeout 600 python tools/witness_ab.py 64 > $O/witness_ab_64.log 2>&1; cat $O/witness_ab_64.log | tail -3
