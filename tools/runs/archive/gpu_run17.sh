#!/bin/bash
# round-1 run 17: ubench with the library products, bench (ALU roofline, 4 producers), N=2 rehearsal (gloo, shared GPU),
# production-size proof (256 tx, 2^24 domain)
set -x
mkdir -p gpurun_out/r17
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
./tools/ubench_int > gpurun_out/r17/ubench.txt 2>&1
BZK_TIMING=1 timeout 600 python bench.py > gpurun_out/r17/bench.txt 2> gpurun_out/r17/bench_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r17/bench_n2_dryrun.txt 2> gpurun_out/r17/bench_n2_err.txt
timeout 900 python tools/prove_production.py 4 3 0 > gpurun_out/r17/production.txt 2> gpurun_out/r17/production_err.txt
echo finished
