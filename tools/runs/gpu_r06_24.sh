#!/bin/bash
# round-6 run 24: G1 accumulation with the next base through LDS by direct loads (shipped build) against the register form (libbzk.so.acclds0): parity, A/B, pipelined ceiling
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run24; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_groth16.py -m gpu -q --timeout=420 --durations=4 -x -k "not 2p26" ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -4 $O/pytest_msm.txt | cut -c1-200
timeout 100 python tests/tools/fuzz_gpu.py 20 2424 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
timeout 1500 python tools/sweep.py r6acclds > $O/acclds_sweep.txt 2>&1
python - <<PY
import json
for l in open("$O/acclds_sweep.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["mode"], d["log_n"], "lib", d.get("lib"), "thr", d.get("throughput"), d.get("mean_ms"), d["ms"], d.get("same_as_raw", d.get("same_as_plain")), {k: v for k, v in d["prof"].items() if "accum" in k})
    else:
        print(l.strip()[:200])
PY
for rep in 1 2; do
for lib in bazuka_amd/libbzk.so.acclds0 ""; do
BZK_LIBBZK=$lib timeout 300 python tools/pipe_probe.py 4 24 > $O/pipe_${rep}_$(basename "$lib" | tr -d .).txt 2>&1; echo "lib=$lib"; tail -1 $O/pipe_${rep}_$(basename "$lib" | tr -d .).txt | cut -c1-400
done
done
echo finished
