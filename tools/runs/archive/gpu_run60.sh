#!/bin/bash
# run 60: N = 2 rehearsal of the final bench.py on the 1-GPU box (ranks share the GPU, 97-byte exchange over gloo: NOT a measurement)
set -x
mkdir -p gpurun_out/r60
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r60/bench_n2_dryrun.txt 2> gpurun_out/r60/bench_n2_err.txt; echo "rc=$?" >> gpurun_out/r60/bench_n2_dryrun.txt
cut -c1-700 gpurun_out/r60/bench_n2_dryrun.txt; tail -3 gpurun_out/r60/bench_n2_err.txt
echo finished
