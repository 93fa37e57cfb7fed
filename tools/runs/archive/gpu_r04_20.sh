#!/bin/bash
# round-4 run 20: cooperative Poseidon for every width <= 8 on small batches: parity, then the latency A/B on the device state and make_work
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_state_device.py tests/test_gpu_state_compress.py tests/test_gpu_mpn_devtree.py tests/test_gpu_mpn_tree.py tests/test_gpu_tree4.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
for v in 0 1; do
  echo "== BZK_POSEIDON_COOP_ANY=$v" >> $O/ab.txt
  BZK_POSEIDON_COOP_ANY=$v timeout 300 python tools/scratch/state_dev_probe.py 2>&1 | tail -1 >> $O/ab.txt
  BZK_POSEIDON_COOP_ANY=$v timeout 300 python - >> $O/ab.txt 2>&1 <<'PY'
import json, bench
from bazuka_amd import Bzk
ctx = Bzk(0)
out = {}
try:
    out = bench.make_work_section(ctx)
except Exception as e:
    out = {"error": repr(e)}
print(json.dumps(out))
PY
done
cut -c1-900 $O/ab.txt
echo finished
