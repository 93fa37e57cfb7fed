#!/bin/bash
set -x
mkdir -p gpurun_out/r53
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --no-proofs --no-cpu-baseline > gpurun_out/r53/bench_msm.txt 2> gpurun_out/r53/err.txt
timeout 300 python bench.py --no-proofs --no-cpu-baseline --log-n 22 --steps 10 > gpurun_out/r53/bench_msm_22.txt 2>> gpurun_out/r53/err.txt
python -c "
import json
for f in ('bench_msm','bench_msm_22'):
    d=json.loads(open('gpurun_out/r53/%s.txt'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('two_msms_in_flight'))"
tail -3 gpurun_out/r53/err.txt
echo finished
