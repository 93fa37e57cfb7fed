#!/bin/bash
# round-4 run 3: endomorphism form of the MSM (GLV on G1, GLS on G2): parity tests, full suite, bench (A/B against BZK_MSM_NO_ENDO=1)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_endo.py -m gpu -q -x > $O/pytest_endo.txt 2>&1; echo "rc=$?" >> $O/pytest_endo.txt
tail -15 $O/pytest_endo.txt
if grep -q "rc=0" $O/pytest_endo.txt; then
  BZK_TEST_PRODUCTION_BYTES=0 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_endo.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
  tail -8 $O/pytest_gpu.txt
  timeout 900 python bench.py > $O/bench_endo.txt 2> $O/bench_endo_err.txt
  BZK_MSM_NO_ENDO=1 timeout 900 python bench.py --no-production > $O/bench_plain.txt 2> $O/bench_plain_err.txt
  python - <<PY
import json
for f in ("bench_endo","bench_plain"):
    try:
        d=json.loads(open("$O/%s.txt"%f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    p=d.get("proofs",{}); o=d.get("other_configs",{})
    print(f, {k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, d["kernel_ms_per_step"])
    print("   proofs", {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring")}, "g2", o.get("msm_g2_2p20",{}).get("ms"), o.get("msm_g2_2p20",{}).get("kernel_ms"))
    pb=o.get("production_block")
    if pb: print("   production", {k:(v.get("prove_s") if isinstance(v,dict) else v) for k,v in pb.items() if k!="what"})
PY
fi
echo finished
