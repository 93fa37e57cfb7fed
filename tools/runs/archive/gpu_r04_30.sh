#!/bin/bash
# round-4 run 30: the G1 accumulation with its eight products inlined, Poseidon's partial rounds inlined, Montgomery quotient digits by v_mad_u64_u32:
# same-box A/B against the previous forms (libbzk.so.base / .inl / .mdsr from tools/build_variant.sh), parity of the default build, kernel trace + PMC
# passes of the headline command on the new sources, default bench line.  Ordered by value: the lease may end before the last steps.
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run30; mkdir -p $O
export TMPDIR=/tmp
timeout 330 python tools/sweep.py r4inl > $O/sweep_inl.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_poseidon_ntt.py tests/test_gpu_tree4.py tests/test_golden_gpu.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
STAMP=$(python -c "import bench; print(bench.msm_source_stamp())")
python tools/pmc_traffic.py $F $W msm_accumulate $O/pmc_traffic.json --calib-from profiles/r04_pmc_traffic.json --stamp $STAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- $CMD" > $O/pmc_traffic.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cat $O/sweep_inl.txt | cut -c1-420; tail -9 $O/pytest_subset.txt; tail -2 $O/smoke.txt; head -8 $O/trace_summary.txt | cut -c1-150; cut -c1-300 $O/pmc_traffic.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"), {k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")})
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20") if k in o})
PY
echo finished
