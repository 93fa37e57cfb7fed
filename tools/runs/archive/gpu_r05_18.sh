#!/bin/bash
# round-5 run 18: level-2 chunk of the prover's two-level bucket reductions (G1: quads of lanes, BZK_MSM_L2_CH; G2: pairs, BZK_MSM_PAIR_L2_CH), 4 (default) against 2:
# one proof at a time and the GPU-side ceiling of the pipelined rate, alternating, same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run18; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for ch in "4 4" "2 2" "2 4"; do set -- $ch; echo "BZK_MSM_L2_CH=$1 BZK_MSM_PAIR_L2_CH=$2"; BZK_MSM_L2_CH=$1 BZK_MSM_PAIR_L2_CH=$2 timeout 100 python tools/prove_serial.py 8 2>&1 | tail -1 | cut -c1-300; BZK_MSM_L2_CH=$1 BZK_MSM_PAIR_L2_CH=$2 timeout 100 python tools/pipe_probe.py 4 16 2>&1 | tail -1 | cut -c1-300; done; done > $O/l2_chunk_ab.txt 2>&1
cat $O/l2_chunk_ab.txt
echo finished
