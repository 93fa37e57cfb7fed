#!/bin/bash
set -x
mkdir -p gpurun_out/r43
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r43/bench_n2_dryrun.txt 2> gpurun_out/r43/bench_n2_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 3 --warmup 1 --no-proofs > gpurun_out/r43/bench_n4_dryrun.txt 2> gpurun_out/r43/bench_n4_err.txt
echo finished
