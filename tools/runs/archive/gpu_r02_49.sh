#!/bin/bash
# round-2 run 49: soak of the final build - 1200 proofs through 4 prover slots, EVERY proof checked with bzk_groth16_verify against its own public inputs
O=gpurun_out/r02_49
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tests/tools/soak.py 1200 4 > $O/soak.txt 2> $O/soak_err.txt; echo "rc=$?" >> $O/soak.txt
cat $O/soak.txt; tail -3 $O/soak_err.txt | cut -c1-300
echo finished
