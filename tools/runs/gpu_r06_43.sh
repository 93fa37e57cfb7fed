#!/bin/bash
# round-6 run 43: runtime settings against the stand-alone headline (one stream, ~25 launches + one 27 KB read-back per call): SDMA off, kernel arguments, blocking waits
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run43; mkdir -p $O
export TMPDIR=/tmp
run() { echo "## $*"; env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --no-proofs --no-others --no-cpu-baseline --no-overlap 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_launch_ms'], round(d['ms_per_step']-sum(d['kernel_ms_per_step'].values()),4))"; }
( for rep in 1 2; do
  run A=0
  run HSA_ENABLE_SDMA=0
  run HIP_FORCE_DEV_KERNARG=0
  run BZK_SYNC_BLOCKING=1
  run GPU_MAX_HW_QUEUES=4
  run HIP_MEM_POOL_SUPPORT=0
done ) > $O/headline_env.txt 2>&1
cat $O/headline_env.txt
echo finished
