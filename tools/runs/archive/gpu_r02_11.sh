#!/bin/bash
# round-2 run 11: 32-bit de-duplication keys, product-side groth16_verify in the worker loop, dense MPN state compress: parity suites,
# serial proof breakdown, production 256-tx proof, bench (with other_configs)
set -x
O=gpurun_out/r02_11
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_worker.py tests/test_gpu_mpn_tree.py tests/test_golden_gpu.py -m gpu -q -x --durations=5 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
BZK_TIMING=1 BZK_PROVE_SERIAL=1 timeout 200 python tools/prove_bench.py 3 > $O/prove_serial.txt 2> $O/prove_serial_err.txt
grep "serial " $O/prove_serial_err.txt | tail -4 > $O/serial_last_proof.txt; grep "groth16_prove:" $O/prove_serial_err.txt | tail -2 >> $O/serial_last_proof.txt; rm -f $O/prove_serial_err.txt
timeout 400 python tests/tools/prove_production.py 4 3 0 > $O/production_256tx.txt 2>/dev/null
timeout 600 python bench.py > $O/bench.txt 2> $O/bench_err.txt
tail -9 $O/pytest.txt; cut -c1-330 $O/serial_last_proof.txt; tail -1 $O/production_256tx.txt | cut -c1-1100; cat $O/bench.txt
echo finished
