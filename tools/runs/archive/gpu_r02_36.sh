#!/bin/bash
# round-2 run 36: production 256-tx proof (2^24 domain) with the h query through a static table (BZK_PROVE_H_TABLE_MAX_LOG=24: 13 levels x 112 B x 2^24 = 24 GB)
set -x
O=gpurun_out/r02_36
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BZK_PROVE_H_TABLE_MAX_LOG=24 timeout 500 python tests/tools/prove_production.py 4 4 0 > $O/production_256tx_htable.txt 2> $O/err1.txt; tail -2 $O/err1.txt
timeout 400 python tests/tools/prove_production.py 4 4 0 > $O/production_256tx.txt 2> $O/err2.txt
for f in production_256tx_htable production_256tx; do tail -1 $O/$f.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['gpu_prove_s'], d.get('pairing_check'))"; done
echo finished
