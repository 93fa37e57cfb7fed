#!/bin/bash
# round-2 run 45: the separate-process producers of run 44 at nice 19, and pinned to the OTHER socket's cores (taskset), beside the 4-slot probe
O=gpurun_out/r02_45
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
rocm-smi --showtopo 2>/dev/null | grep -i "numa\|affinity" | head -6
echo -n "nice 19: "; BZK_PROBE_CHILD_NICE=19 timeout 120 python tools/pipe_probe.py 4 24 8 8 1 2>/dev/null | tail -1
echo -n "prover on cpus 0-63,128-191 (node 0), producers anywhere: "; timeout 120 taskset -c 0-63,128-191 python tools/pipe_probe.py 4 24 8 8 1 2>/dev/null | tail -1
echo -n "prover on node 1 cpus: "; timeout 120 taskset -c 64-127,192-255 python tools/pipe_probe.py 4 24 8 8 1 2>/dev/null | tail -1
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
