#!/bin/bash
# round-1 run 52: state check of the final build: full GPU suite, smoke, default bench, kernel trace + PMC passes of the MSM
# headline (same command), production-size proofs with the faster host generator
set -x
mkdir -p gpurun_out/r52
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r52/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r52/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r52/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r52/smoke.txt
timeout 600 python bench.py > gpurun_out/r52/bench.txt 2> gpurun_out/r52/bench_err.txt
CMD="python bench.py --steps 3 --warmup 1 --no-proofs --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r52/trace -- $CMD > gpurun_out/r52/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r52/pmc_fetch -- $CMD > gpurun_out/r52/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r52/pmc_write -- $CMD > gpurun_out/r52/pmc_write.log 2>&1
F=$(find gpurun_out/r52/pmc_fetch -name "*.db" | head -1); W=$(find gpurun_out/r52/pmc_write -name "*.db" | head -1); T=$(find gpurun_out/r52/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $T > gpurun_out/r52/trace_summary.txt 2>&1
python tools/pmc_traffic.py $F $W msm_accumulate gpurun_out/r52/pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > gpurun_out/r52/pmc_traffic.log 2>&1
find gpurun_out/r52 -name "*.db" -delete; find gpurun_out/r52 -name "*.csv" -size +200k -delete
BZK_DEBUG=1 timeout 300 python tests/tools/prove_production.py 4 4 0 > gpurun_out/r52/production_256tx.txt 2> gpurun_out/r52/production_256tx_err.txt
grep "update_synthesize\|workers\|arrays sized" gpurun_out/r52/production_256tx_err.txt | tail -6 > gpurun_out/r52/production_256tx_host_phases.txt
rm -f gpurun_out/r52/production_256tx_err.txt
timeout 300 python tests/tools/prove_block.py > gpurun_out/r52/production_block.txt 2>&1
tail -3 gpurun_out/r52/pytest_gpu.txt; cat gpurun_out/r52/smoke.txt | tail -2; cut -c1-400 gpurun_out/r52/bench.txt; cat gpurun_out/r52/pmc_traffic.log; head -12 gpurun_out/r52/trace_summary.txt
echo finished
