#!/bin/bash
# round-2 run 20: two-level bucket reduction (BZK_F_THROUGHPUT / BZK_MSM_REDUCE2): parity, single-MSM A/B, proof latency + rate A/B
set -x
O=gpurun_out/r02_20
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_worker.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not tree_2p24 and not ntt_2p24" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
# the whole MSM test file again with the two-level form forced for every call
BZK_MSM_REDUCE2=1 timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -q -x > $O/pytest_forced.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_forced.txt
for r in 0 1; do
  echo "== BZK_MSM_REDUCE2=$r (single 2^20 G1 MSM, G2 in other_configs)"
  BZK_MSM_REDUCE2=$r timeout 300 python bench.py --steps 20 --warmup 3 --no-proofs --no-cpu-baseline --no-overlap 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; g2=d['other_configs']['msm_g2_2p20']; tb=d['other_configs']['msm_g1_2p20_static_table']
print('g1 ms/step', d['ms_per_step'], {x:k[x] for x in k if 'reduce' in x or 'window' in x})
print('g2 ms', g2['ms'], {x:g2['kernel_ms'][x] for x in g2['kernel_ms'] if 'reduce' in x or 'window' in x})
print('table ms', tb['ms'], {x:tb['kernel_ms'][x] for x in tb['kernel_ms'] if 'reduce' in x or 'window' in x})"
done | tee $O/msm_ab.txt
for lat in 0 1; do for s in 1 4; do echo -n "BZK_PROVE_LATENCY=$lat "; BZK_PROVE_LATENCY=$lat timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done; done | tee $O/pipe_probe.txt
for lat in 0 1; do echo "BZK_PROVE_LATENCY=$lat"; BZK_PROVE_LATENCY=$lat BZK_TIMING=1 timeout 200 python tools/prove_bench.py 2>&1 | tail -4 | cut -c1-700; done > $O/prove_single.txt 2>&1
timeout 500 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -3 $O/pytest.txt; tail -3 $O/pytest_forced.txt; cat $O/pipe_probe.txt; cat $O/prove_single.txt; python - <<PY
import json
t=open("$O/bench.txt").read().strip().splitlines()
d=json.loads(t[-1]); p=d["proofs"]
print("msm ms/step", d["ms_per_step"], d["value"], "| gpu_prove_s", p.get("gpu_prove_s"), "pipelined", p.get("proofs_per_s_pipelined"))
PY
tail -3 $O/bench_err.txt
echo finished
