#!/bin/bash
# round-5 run 21: after the last host-side change of witfill.hip (program upload: nothing in flight from locals on a failed upload) - the deferral tests and smoke
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run21; mkdir -p $O
export TMPDIR=/tmp
( time timeout 150 python -m pytest tests/test_gpu_defer.py -m gpu -q -x --timeout=100 ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
echo finished
