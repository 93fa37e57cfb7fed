set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_groth16.py -x -q -m gpu > $O/pytest_ntt_groth16.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ntt_groth16.log
tail -6 $O/pytest_ntt_groth16.log
for v in 0 1 3; do BZK_NTT_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_poseidon_ntt.py -x -q -m gpu -k ntt > $O/pytest_ntt_v$v.log 2>&1; echo "variant $v rc=$?" >> $O/pytest_ntt_v$v.log; tail -2 $O/pytest_ntt_v$v.log; done
timeout 1500 python tools/sweep.py r3ntt > $O/sweep_ntt.log 2>&1
cat $O/sweep_ntt.log
