#!/bin/bash
# round-4 run 31: the shipped build (G1 accumulation and Poseidon partial rounds with inlined products; quotient digit back to v_mul_lo_u32):
# PMC traffic of the headline kernel stamped for these sources, kernel trace of the headline command, default bench line, then the GPU tests that run 30 did not run.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run31; mkdir -p $O
export TMPDIR=/tmp
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib-from profiles/r04_pmc_traffic_run21.json --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
cut -c1-300 $O/pmc_traffic.log; head -8 $O/trace_summary.txt | cut -c1-150
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source"), {k:p.get(k) for k in ("witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")})
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20") if k in o})
PY
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_kernel<bzk::G2Fast:gather" > $O/pmc_other.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cut -c1-900 $O/pmc_other.log
timeout 230 python -m pytest tests/test_gpu_production.py tests/test_gpu_mpn_tree.py tests/test_gpu_mpn_devtree.py tests/test_gpu_state_compress.py tests/test_gpu_state_device.py tests/test_gpu_worker.py tests/test_gpu_mg.py -m gpu -q -x -k "not update_15_3_4" --durations=4 > $O/pytest_rest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.txt
tail -8 $O/pytest_rest.txt
echo finished
