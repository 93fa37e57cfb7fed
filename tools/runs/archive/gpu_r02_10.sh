#!/bin/bash
# round-2 run 10: production sizes on the round-2 build (256-tx Update circuit, 14.4 M constraints, 2^24 domain: pairing-verified; the production
# block = deposit + withdraw + update proofs), prover-slot count of the pipelined bench (3 vs 4 vs 5), PMC passes + trace of the final MSM sources
set -x
O=gpurun_out/r02_10
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python tests/tools/prove_production.py 4 3 0 > $O/production_256tx.txt 2> $O/production_256tx_err.txt; tail -2 $O/production_256tx_err.txt
timeout 300 python tests/tools/prove_block.py > $O/production_block.txt 2>&1
for s in 3 4 5; do
  BZK_BENCH_SLOTS=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-others --no-cpu-baseline --no-overlap > $O/bench_slots$s.txt 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_slots$s.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("slots $s: pipelined", p.get("proofs_per_s_pipelined"), "gpu_prove_s", p.get("gpu_prove_s"), p.get("pipeline"))
PY
done 2>&1 | grep slots | tee $O/slots_summary.txt
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
find $O -name "*.db" -delete; find $O -name "*.csv" -size +300k -delete
cat $O/production_256tx.txt | cut -c1-1200; cat $O/production_block.txt | tail -5 | cut -c1-900; cat $O/slots_summary.txt; cat $O/pmc_traffic.log | cut -c1-600; head -8 $O/trace_summary.txt
echo finished
