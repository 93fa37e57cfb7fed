#!/bin/bash
# round-1 run 27: state check - full GPU suite, bench, N=2 rehearsal, production-size proof with de-duplication
set -x
mkdir -p gpurun_out/r27
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r27/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r27/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r27/bench.txt 2> gpurun_out/r27/bench_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r27/bench_n2_dryrun.txt 2> gpurun_out/r27/bench_n2_err.txt
timeout 900 python tools/prove_production.py 4 3 0 > gpurun_out/r27/production.txt 2> gpurun_out/r27/production_err.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r27/smoke.txt 2>&1
echo finished
