#!/bin/bash
# round-5 run 4 (first lease after the container of runs 1 - 3 was lost with its outputs): the whole GPU suite on the committed build (pair kernels default),
# smoke, same-box A/B of the G2 forms, the per-kernel table of one proof, kernel trace of the headline command, the default bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run4; mkdir -p $O
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 500 python tools/sweep.py r5g2 > $O/g2_forms_ab.txt 2>&1
cut -c1-600 $O/g2_forms_ab.txt
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; echo "BZK_G2_PAIR=$1 BZK_G2_PAIR_TAILS=$2"; BZK_G2_PAIR=$1 BZK_G2_PAIR_TAILS=$2 timeout 200 python tools/pipe_probe.py 4 16 2>&1 | tail -1 | cut -c1-400; done > $O/pipe_probe_ab.txt 2>&1
cat $O/pipe_probe_ab.txt
BZK_PROVE_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/serial_trace -- python tools/prove_serial.py 6 > $O/serial.log 2>&1
T=$(find $O/serial_trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/serial_proof_kernel_table.txt 2>&1
head -40 $O/serial_proof_kernel_table.txt | cut -c1-150
CMD="python bench.py --steps 20 --warmup 3 --no-proofs --no-others --no-overlap --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/trace_summary.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
head -16 $O/trace_summary.txt | cut -c1-150
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; o=d["other_configs"]; pb=o.get("production_block",{})
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"), d.get("kernel_ms_per_step"))
print({k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")})
print({k:(v.get("prove_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
print({k:(o[k].get("ms")) for k in ("tree_2p24","ntt_2p24","h_stage_2p20","msm_g2_2p20","msm_g1_2p24") if k in o})
PY
tail -3 $O/bench_err.txt
echo finished
