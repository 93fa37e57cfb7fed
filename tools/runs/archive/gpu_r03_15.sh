#!/bin/bash
# round-3 run 15: worker tests incl. the multi-slot round
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_worker.py -x -q -m gpu > $O/pytest_worker.txt 2>&1; echo "rc=$?" >> $O/pytest_worker.txt; tail -12 $O/pytest_worker.txt
