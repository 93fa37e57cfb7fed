#!/bin/bash
# round-6 run 11: the paired bit-sum kernel with 256-lane halves (112 KiB of static LDS per workgroup: does the runtime take it, and is it faster?) against 128-lane halves
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run11; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for W in 0 1; do
( BZK_MSM_BITSUM_WIDE=$W timeout 300 python bench.py --steps 20 --warmup 5 --no-proofs --no-others --no-cpu-baseline --no-overlap ) > $O/bench_w${W}_$rep.txt 2> $O/bench_err_w${W}_$rep.txt
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w${W}_$rep.txt").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
    print("wide=$W", {x:d[x] for x in ("value","ms_per_step")}, k["msm_accumulate"], k["msm_bitsum"], k["msm_rowcol"])
except Exception as e:
    print("wide=$W failed", e); print(open("$O/bench_err_w${W}_$rep.txt").read()[-600:])
PY
done
done
BZK_MSM_BITSUM_WIDE=1 timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q --timeout=300 -x -k "not 2p26 and not 2p24" 2>&1 | tail -3
echo finished
