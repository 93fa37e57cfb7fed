#!/bin/bash
# round-2 run 9: rocPRIM radix sorts with an explicit tuned configuration (8 bits per onesweep pass, merge-sort limit lowered) against
# the library's untuned gfx950 fallback (BZK_MSM_SORT_DEFAULT=1): parity (MSM, groth16, proofs), A/B sweep, proof breakdown, bench
set -x
O=gpurun_out/r02_9
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_golden_gpu.py tests/test_gpu_worker.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 600 python tools/sweep.py r2sortcfg > $O/sweep.txt 2>&1
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > $O/prove_serial.txt 2> $O/prove_serial_err.txt
grep "serial " $O/prove_serial_err.txt | tail -4 > $O/serial_last_proof.txt; grep "groth16_prove:" $O/prove_serial_err.txt | tail -2 >> $O/serial_last_proof.txt; rm -f $O/prove_serial_err.txt
timeout 400 python bench.py --no-others > $O/bench.txt 2>/dev/null
tail -4 $O/pytest.txt; cut -c1-420 $O/sweep.txt; cut -c1-700 $O/serial_last_proof.txt; python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("msm ms/step", d["ms_per_step"], d["value"], d["kernel_ms_per_step"], "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"), "2msm", d["two_msms_in_flight"]["value"])
PY
echo finished
