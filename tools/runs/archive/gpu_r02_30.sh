#!/bin/bash
# round-2 run 30: state check of the round's final build: full GPU suite, smoke, fuzz, rehearsal of the N = 2 bench path on one GPU
# (BZK_BENCH_DRYRUN_BACKEND=gloo: NOT a measurement), kernel trace + PMC passes stamped for these MSM sources, default bench
set -x
O=gpurun_out/r02_30
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 200 python tests/tools/fuzz_gpu.py 60 8 > $O/fuzz.txt 2>&1
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $O/dryrun_weak_n2.txt 2> $O/dryrun_weak_n2_err.txt; echo "dryrun weak rc=$?"
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-proofs --scaling strong --log-n-total 22 > $O/dryrun_strong_n2.txt 2> $O/dryrun_strong_n2_err.txt; echo "dryrun strong rc=$?"
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline --no-others --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_calib -- ./tools/ubench_batched_affine calib > $O/pmc_calib.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1); T=$(find $O/trace -name "*.db" | head -1); C=$(find $O/pmc_calib -name "*.db" | head -1)
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
REQ=$(grep "calib gather" $O/pmc_calib.log | head -1 | sed 's/.*requested \([0-9]*\) bytes.*/\1/')
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib $C $REQ --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
grep "calib gather" $O/pmc_calib.log > $O/pmc_calib_lines.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +300k -delete
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json
timeout 600 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -9 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; tail -1 $O/fuzz.txt | cut -c1-300; tail -1 $O/dryrun_weak_n2.txt | cut -c1-1200; tail -3 $O/dryrun_weak_n2_err.txt | cut -c1-300; tail -1 $O/dryrun_strong_n2.txt | cut -c1-700; cut -c1-300 $O/pmc_traffic.log; head -10 $O/trace_summary.txt; cat $O/bench.txt
echo finished
