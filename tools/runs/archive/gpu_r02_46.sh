#!/bin/bash
# round-2 run 46: the GPU box's container has a CPU quota (cgroup cpu.max = 16 CPUs of the 256 it shows) - how much of it do the prover's own
# spinning waits take?  probe and bench pipeline with blocking waits (BZK_SYNC_BLOCKING=1) against the default
O=gpurun_out/r02_46
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
stat() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
{
cat /sys/fs/cgroup/cpu.max
for b in 0 1; do
  echo "== BZK_SYNC_BLOCKING=$b"; stat
  echo -n "probe 4 slots: "; BZK_SYNC_BLOCKING=$b timeout 120 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1; stat
  echo -n "probe 4 slots + 8x8 producers (threads): "; BZK_SYNC_BLOCKING=$b timeout 120 python tools/pipe_probe.py 4 24 8 8 0 2>/dev/null | tail -1; stat
  echo -n "bench pipeline: "; BZK_SYNC_BLOCKING=$b timeout 200 python bench.py --steps 3 --warmup 1 --no-others --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['proofs']
print('msm ms/step', d['ms_per_step'], 'pipelined', p.get('proofs_per_s_pipelined'), 'synth_under_load', p.get('producer_synth_s_mean_under_load'), 'gpu_prove_s', p.get('gpu_prove_s'))"; stat
done
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
