#!/bin/bash
# round-2 run 24: full GPU suite + fuzz on the G2 static-bound mixed addition (after the fix for bases that are Fp2 product outputs)
set -x
O=gpurun_out/r02_24
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 60 5 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-400
timeout 200 python tests/tools/fuzz_gpu.py 40 6 > $O/fuzz2.txt 2>&1; tail -1 $O/fuzz2.txt | cut -c1-400
tail -4 $O/pytest.txt
echo finished
