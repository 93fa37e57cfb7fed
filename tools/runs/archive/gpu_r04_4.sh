#!/bin/bash
# round-4 run 4: where the endomorphism form pays - per curve, stand-alone and inside pipelined proofs
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/sweep.py r4endo > $O/sweep_endo.txt 2>&1
for cfg in "0 0" "0 1" "2 1" "1 1"; do
  set -- $cfg
  BZK_MSM_ENDO_G1=$1 BZK_MSM_ENDO_G2=$2 BZK_BENCH_TWO_PROCS=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-others --no-overlap --no-cpu-baseline > $O/bench_g1_$1_g2_$2.txt 2> $O/bench_g1_$1_g2_$2.err
done
cut -c1-900 $O/sweep_endo.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_g1_*.txt")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); p=d["proofs"]
        print(f.split("/")[-1], d["value"], {k:p.get(k) for k in ("gpu_prove_s","proofs_per_s_pipelined","proofs_per_s_ring","proofs_per_s_serial")})
    except Exception as e:
        print(f, "failed", e)
PY
echo finished
