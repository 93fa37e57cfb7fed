#!/bin/bash
# run 46: folded static-base tables (tests + A/B), worker loop end-to-end, full GPU suite
set -x
mkdir -p gpurun_out/r46
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r46/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r46/pytest.txt
tail -5 gpurun_out/r46/pytest.txt
timeout 600 python tools/sweep.py fold > gpurun_out/r46/fold.txt 2>&1
cat gpurun_out/r46/fold.txt
echo finished
