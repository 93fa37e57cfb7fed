#!/bin/bash
# round-2 run 8: state check - full GPU suite, smoke, default bench (dominant-kernel-only events inside the timed region), rehearsal of the
# N = 2 code paths (weak and --scaling strong) on one GPU with BZK_BENCH_DRYRUN_BACKEND=gloo (NOT a measurement)
set -x
O=gpurun_out/r02_8
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 600 python bench.py > $O/bench.txt 2> $O/bench_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-proofs > $O/dryrun_weak_n2.txt 2> $O/dryrun_weak_n2_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-proofs --scaling strong --log-n-total 22 > $O/dryrun_strong_n2.txt 2> $O/dryrun_strong_n2_err.txt
tail -16 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; cat $O/bench.txt; cut -c1-900 $O/dryrun_weak_n2.txt; cut -c1-900 $O/dryrun_strong_n2.txt; tail -3 $O/dryrun_strong_n2_err.txt
echo finished
