#!/bin/bash
# round-2 run 4: bucket-reduction chunk size in THROUGHPUT mode (pipelined proofs): the per-chunk scalar multiplication is 60 % of the
# reduction's work at ch = 8; larger chunks trade chain length (single-MSM latency) for work.  Also: bellman params round trip test.
set -x
mkdir -p gpurun_out/r02_4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_groth16.py -m gpu -q -x > gpurun_out/r02_4/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_4/pytest.txt
for ch in 8 16 32 64; do
  BZK_MSM_CHUNK=$ch timeout 300 python bench.py --steps 5 --warmup 2 --no-others --no-cpu-baseline --no-overlap > gpurun_out/r02_4/bench_ch$ch.txt 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r02_4/bench_ch$ch.txt").read().strip().splitlines()[-1])
p=d["proofs"]
print("chunk $ch: msm ms/step", d["ms_per_step"], "reduce", d["kernel_ms_per_step"].get("msm_reduce"), "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"), "gpu_only", p.get("proofs_per_s_gpu_only"))
PY
done 2>&1 | tee gpurun_out/r02_4/summary.txt
tail -3 gpurun_out/r02_4/pytest.txt
