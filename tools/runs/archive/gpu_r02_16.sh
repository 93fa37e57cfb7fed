#!/bin/bash
# round-2 run 16: msm_reduce<G1> in memory-operand form (256 instead of 354 registers: two waves per SIMD, can share a SIMD with an accumulate wave):
# parity, single-MSM sweep, proof breakdown, pipelined proofs, probe
set -x
O=gpurun_out/r02_16
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_golden_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 300 python tools/sweep.py r35 > $O/sweep.txt 2>&1
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > $O/prove_serial.txt 2> $O/prove_serial_err.txt
grep "serial " $O/prove_serial_err.txt | tail -4 > $O/serial_last_proof.txt; grep "groth16_prove:" $O/prove_serial_err.txt | tail -2 >> $O/serial_last_proof.txt; rm -f $O/prove_serial_err.txt
for s in 1 4; do timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done | tee $O/pipe_probe.txt
timeout 400 python bench.py --no-others --no-cpu-baseline > $O/bench.txt 2>/dev/null
tail -3 $O/pytest.txt; cut -c1-330 $O/sweep.txt; cut -c1-700 $O/serial_last_proof.txt | grep -o "serial [a-z_0-9]*\|msm_reduce 1 [0-9.]*\|msm_accumulate 1 [0-9.]*\|lanes started [0-9.]*"; cat $O/pipe_probe.txt; python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("msm ms/step", d["ms_per_step"], d["value"], "reduce", d["kernel_ms_per_step"]["msm_reduce"], "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"), "2msm", d["two_msms_in_flight"]["value"])
PY
echo finished
