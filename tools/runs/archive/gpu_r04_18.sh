#!/bin/bash
# round-4 run 18: the persistent device state (bzk_state_*): its GPU tests, then the bench entry
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_state_device.py tests/test_gpu_state_compress.py -x -q --durations=8 > $O/pytest_state.txt 2>&1; echo "rc=$?" >> $O/pytest_state.txt
tail -30 $O/pytest_state.txt
timeout 600 python - > $O/bench_state.txt 2>&1 <<'PY'
import json, bench
from bazuka_amd import Bzk
ctx = Bzk(0)
out = {}
bench.state_seam_section(ctx, out)
print(json.dumps(out))
PY
cat $O/bench_state.txt | cut -c1-1500
echo finished
