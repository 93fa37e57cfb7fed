#!/bin/bash
# round-2 run 37: does the mere presence of host witness producers cost the prover?  pipe probe (same witness) with background producers whose output is discarded
set -x
O=gpurun_out/r02_37
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "0 8" "8 8" "6 16" "0 8" "16 8"; do
  set -- $cfg
  timeout 200 python tools/pipe_probe.py 4 24 $1 $2 2>/dev/null | tail -1
done | tee $O/probe_bg.txt
echo finished
