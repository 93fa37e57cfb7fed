#!/bin/bash
# round-5 run 11: state check of the shipped build - the whole GPU suite (every test under its own timeout), smoke, PMC traffic of the headline kernel and of the
# other kernels stamped for these sources, NTT counters, kernel trace of the headline command, the default bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run11; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1000 python -m pytest tests -m gpu -q -p pytest_timeout --timeout=420 --durations=8 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib-from profiles/r04_pmc_traffic_run21.json --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
cut -c1-300 $O/pmc_traffic.log
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
head -14 $O/trace_summary.txt | cut -c1-150
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_g2pair_kernel:gather" > $O/pmc_other.log 2>&1
cut -c1-700 $O/pmc_other.log
# NTT at 2^24: where the wave cycles go (separate passes: the counters do not share slots)
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $O/ntt_c1 -- python tools/pmc_ops.py > $O/ntt_c1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/ntt_c2 -- python tools/pmc_ops.py > $O/ntt_c2.log 2>&1
C1=$(find $O/ntt_c1 -name "*.db" | head -1); C2=$(find $O/ntt_c2 -name "*.db" | head -1)
python tools/pmc_generic.py $C1 SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --ratio SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES --top 12 > $O/ntt_counters.txt 2>&1
python tools/pmc_generic.py $C2 SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --ratio SQ_LDS_BANK_CONFLICT/SQ_ACTIVE_INST_LDS --top 12 >> $O/ntt_counters.txt 2>&1
cut -c1-220 $O/ntt_counters.txt | head -30
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json; cp $O/pmc_other_kernels.json profiles/r05_pmc_other_kernels.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"], d.get("kernel_ms_per_step"), d.get("cpu_baseline",{}).get("value"))
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring","prover_host_cpu_s_per_proof")}, p.get("deferred"), p.get("two_processes",{}).get("proofs_per_s"), p.get("cpu_baseline"))
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms"), o[k].get("roofline",{}).get("traffic")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p24") if k in o})
PY
tail -3 $O/bench_err.txt | cut -c1-300
echo finished
