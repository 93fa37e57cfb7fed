#!/bin/bash
# round-2 run 28: NTT with two butterfly stages per LDS round trip (radix 4 in registers) and no multiplication in stage 0; merged
# count / population-key kernel of the bucket phase: parity + timings (compare r02_run26: ntt 2^20 0.199 ms, 2^24 3.107 ms, h stage 1.36 ms)
set -x
O=gpurun_out/r02_28
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not tree_2p24" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 30 7 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-400
timeout 500 python bench.py > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; oc=d["other_configs"]
print("msm ms/step", d["ms_per_step"], d["value"], "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"))
print({k: oc[k]["ms"] for k in ("ntt_2p20","ntt_2p24","h_stage_2p20")})
print({k: d["kernel_ms_per_step"][k] for k in d["kernel_ms_per_step"]})
PY
tail -3 $O/pytest.txt
echo finished
