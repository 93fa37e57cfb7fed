#!/bin/bash
# round-2 run 7: state of the build after the fold regime fix: MSM parity + size sweep, the batched-affine micro-benchmark (VERDICT r1
# item 6), rocprofv3 kernel trace of the bench command, PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes) with the FETCH_SIZE
# calibration on the kernel's own gather pattern, default bench
set -x
O=gpurun_out/r02_7
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 300 python tools/sweep.py r35 > $O/sweep_sizes.txt 2>&1
timeout 120 ./tools/ubench_batched_affine > $O/ubench_batched_affine.txt 2>&1
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline --no-others --no-overlap"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_calib -- ./tools/ubench_batched_affine calib > $O/pmc_calib.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1); T=$(find $O/trace -name "*.db" | head -1); C=$(find $O/pmc_calib -name "*.db" | head -1)
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O/trace -name "*stats*.csv" | head -3 | while read f; do cp $f $O/$(basename $f); done
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
REQ=$(grep "calib gather" $O/pmc_calib.log | head -1 | sed 's/.*requested \([0-9]*\) bytes.*/\1/')
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib $C $REQ --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
grep "calib gather" $O/pmc_calib.log > $O/pmc_calib_lines.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +300k -delete
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json 2>/dev/null
timeout 600 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -3 $O/pytest.txt; cat $O/sweep_sizes.txt | cut -c1-400; cat $O/ubench_batched_affine.txt; cat $O/pmc_traffic.log; cat $O/pmc_calib_lines.txt; head -14 $O/trace_summary.txt; cut -c1-700 $O/bench.txt
echo finished
