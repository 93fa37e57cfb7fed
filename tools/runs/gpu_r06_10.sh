#!/bin/bash
# round-6 run 10: adjacent bit terms combined on the device (D_j = S_2j + 2 S_(2j+1): half the host additions of the Horner): parity + the headline command twice
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run10; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_endo.py tests/test_gpu_fullsize.py tests/test_gpu_mpn_prove.py tests/test_gpu_groth16.py tests/test_gpu_mg.py -m gpu -q --timeout=420 --durations=4 ) > $O/pytest_msm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_msm.txt
tail -8 $O/pytest_msm.txt | cut -c1-200
timeout 200 python tests/tools/fuzz_gpu.py 30 777 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
for rep in 1 2; do
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-proofs --no-others --no-cpu-baseline ) > $O/bench_headline_$rep.txt 2> $O/bench_headline_err_$rep.txt
python - <<PY
import json
d=json.loads(open("$O/bench_headline_$rep.txt").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_launch_ms"], d.get("kernel_ms_per_step"), d.get("two_msms_in_flight"))
PY
done
timeout 300 python tools/pipe_probe.py 4 24 > $O/pipe.txt 2>&1; tail -1 $O/pipe.txt
echo finished
