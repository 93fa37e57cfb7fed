#!/bin/bash
set -x
mkdir -p gpurun_out/r42
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python tools/prove_block.py > gpurun_out/r42/block.txt 2> gpurun_out/r42/block_err.txt
timeout 300 python -m pytest tests/test_gpu_mpn_prove.py -q > gpurun_out/r42/pytest.txt 2>&1
echo finished
