#!/bin/bash
# round-5 run 24: the deferral / staging / worker tests and smoke on the final library (quota-aware generator threads)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run24; mkdir -p $O
export TMPDIR=/tmp
( time timeout 200 python -m pytest tests/test_gpu_defer.py tests/test_gpu_worker.py tests/test_gpu_mpn_prove.py -m gpu -q -x --timeout=100 ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
echo finished
