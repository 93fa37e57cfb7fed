#!/bin/bash
# round-4 run 7: final state check: full GPU suite, differential fuzzing incl. the endomorphism form, stamped PMC passes (headline kernel +
# NTT / tree / G2 accumulate), default bench, BASELINE configs[2] at face value (--with-1024tx), 4-rank rehearsal
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run7; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
timeout 400 python tests/tools/fuzz_gpu.py 150 41 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_calib -- ./tools/ubench_batched_affine calib > $O/pmc_calib.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1); T=$(find $O/trace -name "*.db" | head -1); C=$(find $O/pmc_calib -name "*.db" | head -1)
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
REQ=$(grep "calib gather" $O/pmc_calib.log | head -1 | sed 's/.*requested \([0-9]*\) bytes.*/\1/')
python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
python tools/rocpd_summary.py $OF > $O/pmc_ops_fetch_summary.txt 2>&1
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib $C $REQ --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_kernel<bzk::G2Fast:gather" > $O/pmc_other.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json; cp $O/pmc_other_kernels.json profiles/r04_pmc_other_kernels.json
( time timeout 1200 python bench.py ) > $O/bench.txt 2> $O/bench_err.txt
( time timeout 1500 python bench.py --steps 5 --warmup 2 --no-proofs --no-overlap --no-cpu-baseline --with-1024tx ) > $O/bench_1024tx.txt 2> $O/bench_1024tx_err.txt
BZK_BENCH_DRYRUN_BACKEND=gloo timeout 900 python bench.py --gpus 4 --steps 10 --warmup 2 > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
tail -10 $O/pytest_gpu.txt; tail -2 $O/smoke.txt; tail -3 $O/fuzz.txt | cut -c1-600; cat $O/pmc_traffic.log | cut -c1-400; cat $O/pmc_other.log | cut -c1-1800; head -10 $O/trace_summary.txt
python - <<PY
import json
for f in ("bench","bench_1024tx"):
    try:
        d=json.loads(open("$O/%s.txt"%f).read().strip().splitlines()[-1])
        o=d.get("other_configs",{}); pb=o.get("production_block",{})
        print(f, {k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"].get("traffic_source"), {k:(v.get("prove_s"),v.get("decode_and_witness_s")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
        print("   traffic:", {k:(o[k]["roofline"].get("traffic"), ) for k in ("tree_2p24","ntt_2p24","msm_g2_2p20") if k in o and "roofline" in o[k]})
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench_1024tx_err.txt; cut -c1-300 $O/bench_dryrun_gpus4.txt
echo finished
