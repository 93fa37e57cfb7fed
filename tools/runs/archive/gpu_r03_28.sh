#!/bin/bash
# round-3 run 28: the last build of the round: full GPU suite, smoke, differential fuzzing, soak (every proof verified), production block
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run28; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 200 python tests/tools/fuzz_gpu.py 120 28 > $O/fuzz.txt 2>&1; echo "rc=$?" >> $O/fuzz.txt; tail -2 $O/fuzz.txt | cut -c1-600
timeout 200 python tests/tools/soak.py 400 4 > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt; tail -2 $O/soak.txt | cut -c1-600
timeout 300 python tests/tools/prove_block.py > $O/production_block.txt 2>&1; echo "rc=$?" >> $O/production_block.txt; tail -2 $O/production_block.txt | cut -c1-900
