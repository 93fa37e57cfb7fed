#!/bin/bash
# round-2 run 5: binary-GCD inversion (fp28::inv) in dedup_affine / setup / to-affine, tree fold for buckets of >= 3 tasks, K = 8:
# parity of everything that inverts or folds, serial per-MSM breakdown of a proof, proof latency and pipelined rate
set -x
mkdir -p gpurun_out/r02_5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py > gpurun_out/r02_5/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_5/pytest.txt
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > gpurun_out/r02_5/prove_serial.txt 2> gpurun_out/r02_5/prove_serial_err.txt
grep "serial " gpurun_out/r02_5/prove_serial_err.txt | tail -4 > gpurun_out/r02_5/serial_last_proof.txt
grep "groth16_prove:" gpurun_out/r02_5/prove_serial_err.txt | tail -2 >> gpurun_out/r02_5/serial_last_proof.txt
rm -f gpurun_out/r02_5/prove_serial_err.txt
timeout 300 python bench.py --no-others --no-cpu-baseline > gpurun_out/r02_5/bench.txt 2>/dev/null
tail -4 gpurun_out/r02_5/pytest.txt; cat gpurun_out/r02_5/serial_last_proof.txt; python - <<PY
import json
d=json.loads(open("gpurun_out/r02_5/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]
print("msm ms/step", d["ms_per_step"], d["value"], "| crs", p.get("gpu_crs_setup_s"), "gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"), "gpu_only", p.get("proofs_per_s_gpu_only"))
PY
