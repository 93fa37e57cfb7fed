#!/bin/bash
# round-4 run 28: the host witness generator on AVX-512 IFMA - the GPU suites that prove what it emits, then the default bench (witness_cpu_s, proofs/s)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run28; mkdir -p $O
grep -m1 -o "avx512ifma" /proc/cpuinfo > $O/cpu.txt; grep -m1 "model name" /proc/cpuinfo >> $O/cpu.txt
timeout 900 python -m pytest tests/test_gpu_mpn_prove.py tests/test_gpu_mpn_devtree.py tests/test_gpu_worker.py tests/test_gpu_fullsize.py -x -q --durations=5 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
for v in 0 1; do
  BZK_HOST_IFMA=$v BZK_BENCH_TWO_PROCS=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline --no-production 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['proofs']
print('BZK_HOST_IFMA=$v', {k:p.get(k) for k in ('witness_s','witness_cpu_s','gpu_prove_s','proofs_per_s_serial','proofs_per_s_pipelined','proofs_per_s_ring','producer_synth_s_mean_under_load')})" >> $O/ab.txt
done
cat $O/cpu.txt $O/ab.txt
( time timeout 1200 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench_err.txt
python - <<PY
import json
d=json.loads(open("$O/bench.txt").read().strip().splitlines()[-1]); p=d["proofs"]; pb=d["other_configs"]["production_block"]
print({k:d[k] for k in ("value","ms_per_step","proofs_per_sec")}, {k:p.get(k) for k in ("witness_s","witness_cpu_s","gpu_prove_s","proofs_per_s_serial","proofs_per_s_pipelined","proofs_per_s_ring")}, p["two_processes"]["proofs_per_s"])
print({k:(v.get("prove_s"),v.get("decode_and_witness_s"),v.get("verified")) if isinstance(v,dict) else v for k,v in pb.items() if k!="what"})
PY
echo finished
