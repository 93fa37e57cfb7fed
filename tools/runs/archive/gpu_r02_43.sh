#!/bin/bash
# round-2 run 43: rehearsal of bench.py --partition points (N = 2 on ONE GPU over gloo: NOT a measurement - it shows the code path runs and
# that the point-sharded fold equals the single-GPU MSM), weak and strong; the default window partition once more beside it
O=gpurun_out/r02_43
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { BZK_BENCH_DRYRUN_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 5 --warmup 2 --no-proofs "${@:2}" 2>$O/err_$1.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['parallelism'], '|', d['scaling'], '| ms/step', d['ms_per_step'], '| windows', d['config']['window_range_this_rank'], 'points', d['config']['point_range_this_rank'], '|', d.get('dryrun'))"; }
{
run 29521 --partition points
run 29522 --partition windows
run 29523 --partition points --scaling strong --log-n-total 22
run 29524 --partition windows --scaling strong --log-n-total 22
} > $O/out.txt 2>&1
cat $O/out.txt; tail -2 $O/err_29521.txt | cut -c1-300
echo finished
