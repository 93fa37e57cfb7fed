#!/bin/bash
# round-4 run 1: production-shape proofs (row g) incl. the opt-in 2^24 byte comparison, then the full GPU suite and the default bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run1; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/host.txt; free -g >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1
BZK_TEST_PRODUCTION_BYTES=1 timeout 1500 python -m pytest tests/test_gpu_production.py -m gpu -q --durations=5 > $O/pytest_production_bytes.txt 2>&1; echo "rc=$?" >> $O/pytest_production_bytes.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -12 $O/pytest_production_bytes.txt; tail -25 $O/pytest_gpu.txt; cut -c1-1500 $O/bench.txt
echo finished
