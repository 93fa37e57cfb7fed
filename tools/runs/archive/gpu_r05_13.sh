#!/bin/bash
# round-5 run 13: the production block (Deposit 2^21 + Withdraw 2^22 + Update 2^24) with the Update work once on the plain and once on the deferred generator
# (256 transitions: the deferred-value program has 256 workgroups to run), every proof verified by the host verifier
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run13; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python - > $O/production_block.txt 2> $O/err.txt <<'PY'
import json, torch, bench
from bazuka_amd import Bzk
torch.cuda.set_device(0)
ctx = Bzk(0)
print(json.dumps(bench.production_block_section(ctx)))
PY
python - <<PY
import json
d = json.loads(open("$O/production_block.txt").read().strip().splitlines()[-1])
for k, v in d.items():
    if isinstance(v, dict): print(k, {x: v.get(x) for x in ("n_constraints", "make_work_s", "decode_and_witness_s", "prove_s", "verified", "verify_ms_host", "deferred")})
    else: print(k, v)
PY
tail -3 $O/err.txt | cut -c1-300
echo finished
