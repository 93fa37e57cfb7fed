#!/bin/bash
set -x
mkdir -p gpurun_out/r29
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python tools/prove_block.py > gpurun_out/r29/block.txt 2> gpurun_out/r29/block_err.txt
echo finished
