#!/bin/bash
# round-5 run 19: LAST state check - the whole GPU suite and smoke on the final commit's library (deferral in all three circuits, workers with --defer, staging)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run19; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1000 python -m pytest tests -m gpu -q --timeout=420 --durations=6 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
echo finished
