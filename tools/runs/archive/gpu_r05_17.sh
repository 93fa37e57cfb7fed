#!/bin/bash
# round-5 run 17: BASELINE configs[2] as ONE circuit - the single 1024-transaction Update circuit (57.8 M constraints, 2^26 domain) with the round-5 kernels, plain
# and deferred generator, inside bench.production_block_section (every proof verified by the host verifier)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run17; mkdir -p $O
export TMPDIR=/tmp
free -g | head -2
( time timeout 560 python - > $O/production_block_1024.txt 2> $O/err.txt <<'PY'
import json, torch, bench
from bazuka_amd import Bzk
torch.cuda.set_device(0)
ctx = Bzk(0)
print(json.dumps(bench.production_block_section(ctx, with_1024tx=True)))
PY
) 2>&1 | tail -3
python - <<PY
import json
d = json.loads(open("$O/production_block_1024.txt").read().strip().splitlines()[-1])
for k, v in d.items():
    if isinstance(v, dict): print(k, {x: v.get(x) for x in ("n_constraints", "log_domain", "gpu_crs_setup_s", "make_work_s", "decode_and_witness_s", "prove_s", "prove_s_all", "verified", "deferred")})
    else: print(k, str(v)[:120])
PY
tail -3 $O/err.txt | cut -c1-300
echo finished
