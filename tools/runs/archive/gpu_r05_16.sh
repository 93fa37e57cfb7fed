#!/bin/bash
# round-5 run 16: the workers with --defer (python and native), the defer tests once more on the last library, smoke
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run16; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_worker.py tests/test_gpu_defer.py -m gpu -q -x --timeout=200 --durations=4 ) > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -12 $O/pytest.txt | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
echo finished
