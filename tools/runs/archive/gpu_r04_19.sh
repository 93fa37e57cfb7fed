#!/bin/bash
# round-4 run 19: the native worker (bzk-worker) against the mock node: its GPU tests + the production block through it
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_worker.py -x -q --durations=8 > $O/pytest_worker.txt 2>&1; echo "rc=$?" >> $O/pytest_worker.txt
tail -25 $O/pytest_worker.txt
echo finished
