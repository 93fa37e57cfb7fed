#!/bin/bash
# round-2 run 17: static-base tables on the round-2 build: full tables (one bucket set, no host Horner) at window sizes 16 .. 21
set -x
O=gpurun_out/r02_17
mkdir -p $O
cd $GRAFT_REPO_ROOT
python - > $O/tables.txt 2>&1 <<'PY'
import subprocess, sys, os
for c in (16, 18, 19, 20, 21):
    e = dict(os.environ, BZK_MSM_TABLE_C=str(c))
    out = subprocess.run([sys.executable, "tools/sweep.py", "child", "g1tab", "20"], env=e, capture_output=True, text=True, timeout=200)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(lines[-1] if lines else "FAILED " + out.stderr[-300:], flush=True)
out = subprocess.run([sys.executable, "tools/sweep.py", "child", "g1", "20"], capture_output=True, text=True, timeout=200)
print([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
PY
cat $O/tables.txt | cut -c1-500
echo finished
