#!/bin/bash
# round-6 run 36: FINAL STATE CHECK (run 32 + the task-cut rule of run 35 + the fuzzer kind ranges):
# the whole GPU suite, smoke, the fuzzer, PMC traffic of the headline kernel and of the other kernels stamped for these sources, kernel trace of the headline command, the default bench line, the 4-rank rehearsal
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run36; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=420 --durations=10 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -18 $O/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 200 python tests/tools/fuzz_gpu.py 40 2525 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt; tail -2 $O/fuzz.txt | cut -c1-400
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib-from profiles/r05_pmc_traffic.json --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
cut -c1-300 $O/pmc_traffic.log
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
head -16 $O/trace_summary.txt | cut -c1-150
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_g2pair_kernel:gather" > $O/pmc_other.log 2>&1
cut -c1-900 $O/pmc_other.log
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json; cp $O/pmc_other_kernels.json profiles/r06_pmc_other_kernels.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"], d.get("kernel_ms_per_step"), d.get("two_msms_in_flight"), d.get("cpu_baseline",{}).get("value"))
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("deferred"), p.get("two_processes",{}).get("proofs_per_s"), p.get("cpu_baseline"))
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms"), o[k].get("roofline",{}).get("traffic")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p24","msm_g1_2p19","msm_g1_2p18","msm_g1_2p20_static_table") if k in o})
PY
tail -3 $O/bench_err.txt | cut -c1-300
( time BZK_BENCH_DRYRUN_BACKEND=gloo timeout 700 python bench.py --gpus 4 --steps 10 --warmup 2 ) > $O/bench_dryrun_gpus4.txt 2> $O/bench_dryrun_gpus4_err.txt
python - <<PY
import json
lines = [l for l in open("$O/bench_dryrun_gpus4.txt").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "proofs_per_sec")})
print(json.dumps(d.get("proofs", {}).get("host_bound"))[:900])
PY
tail -3 $O/bench_dryrun_gpus4_err.txt | cut -c1-300
echo finished
