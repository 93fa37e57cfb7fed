#!/bin/bash
# round-2 run 42: full GPU suite, smoke and default bench on the round's last build (host pool change, 8 x 8 producers)
O=gpurun_out/r02_42
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 400 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; cat $O/bench.txt | cut -c1-1500
echo finished
