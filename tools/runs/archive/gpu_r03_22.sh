#!/bin/bash
# round-3 run 22: state check of the final build: full GPU suite, smoke, default bench, kernel trace + PMC passes of the MSM headline
# (same command), production-size proof with the device witness builder, dry-run of the N = 2 launch path
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run22; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 900 python bench.py > $O/bench.txt 2> $O/bench_err.txt
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_calib -- ./tools/ubench_batched_affine calib > $O/pmc_calib.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1); T=$(find $O/trace -name "*.db" | head -1); C=$(find $O/pmc_calib -name "*.db" | head -1)
REQ=$(grep "calib gather" $O/pmc_calib.log | head -1 | sed 's/.*requested \([0-9]*\) bytes.*/\1/')
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib $C $REQ --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
timeout 400 python tests/tools/prove_production.py 4 4 0 1 > $O/production_256tx_device_builder.txt 2>&1
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-proofs > $O/bench_dryrun_gpus2.txt 2>&1
tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; cut -c1-600 $O/bench.txt; cat $O/pmc_traffic.log; head -14 $O/trace_summary.txt; tail -1 $O/production_256tx_device_builder.txt | cut -c1-700; tail -2 $O/bench_dryrun_gpus2.txt | cut -c1-500
echo finished
