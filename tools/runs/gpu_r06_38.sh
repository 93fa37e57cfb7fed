#!/bin/bash
# round-6 run 38: the runtime's helper threads cost 0.012 CPU-s per proof (mostly system time) whatever the wait mode (run 37): which runtime setting moves it?
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run38; mkdir -p $O
export TMPDIR=/tmp
run() { echo "## $*"; env BZK_SYNC_BLOCKING=1 "$@" timeout 300 python tools/host_cpu_probe.py 4 24 2>&1 | grep '^{'; }
( run A=0
  run HSA_ENABLE_INTERRUPT=0
  run AMD_DIRECT_DISPATCH=0
  run GPU_MAX_HW_QUEUES=4
  run HIP_FORCE_DEV_KERNARG=0
  run HSA_ENABLE_SDMA=0
  run A=0 ) > $O/host_cpu_env.txt 2>&1
cat $O/host_cpu_env.txt
echo finished
