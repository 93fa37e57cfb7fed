#!/bin/bash
# run 51: host generator without the zero fill of the witness arrays + parallel z copy: witness times, full GPU suite
set -x
mkdir -p gpurun_out/r51
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r51/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r51/pytest.txt
tail -3 gpurun_out/r51/pytest.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r51/bench.txt 2> gpurun_out/r51/bench_err.txt
cut -c1-300 gpurun_out/r51/bench.txt; python -c "
import json; d=json.loads(open('gpurun_out/r51/bench.txt').read().strip().splitlines()[-1]); print(json.dumps(d['proofs']))"
BZK_DEBUG=1 timeout 300 python tests/tools/prove_production.py 4 3 0 > gpurun_out/r51/production_256tx.txt 2> gpurun_out/r51/production_256tx_err.txt
cut -c1-420 gpurun_out/r51/production_256tx.txt | tail -1; grep "update_synthesize\|workers" gpurun_out/r51/production_256tx_err.txt | tail -4
echo finished
