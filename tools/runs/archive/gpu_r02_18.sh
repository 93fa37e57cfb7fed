#!/bin/bash
# round-2 run 18: h-query static table inside bzk_groth16_prove (c = 20, 13 levels): parity (MSM tables, proofs at 2^16 / 2^17 / 2^20), proof latency and
# rate with / without it, bench
set -x
O=gpurun_out/r02_18
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_worker.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not tree_2p24 and not ntt_2p24" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
for t in 1 0; do for s in 1 4; do echo -n "h_table=$t "; BZK_PROVE_H_TABLE=$t timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done; done | tee $O/pipe_probe.txt
timeout 500 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -3 $O/pytest.txt; cat $O/pipe_probe.txt; python - <<PY
import json
t=open("$O/bench.txt").read().strip().splitlines()
d=json.loads(t[-1]); p=d["proofs"]
print("msm ms/step", d["ms_per_step"], d["value"], "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"))
print(d["other_configs"].get("msm_g1_2p20_static_table"))
PY
tail -3 $O/bench_err.txt
echo finished
