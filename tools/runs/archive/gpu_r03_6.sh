set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_groth16.py tests/test_gpu_msm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 1200 python tools/sweep.py r3ntt2 > $O/sweep_ntt3.log 2>&1
cat $O/sweep_ntt3.log
timeout 600 python bench.py --no-proofs --no-others --no-cpu-baseline > $O/bench_msm.log 2>&1; tail -c 1800 $O/bench_msm.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/tools/sweep.py child ntt 20 > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$O/pmc -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/pmc_summary.txt 2>&1
grep -i "ntt_pass" $GRAFT_REPO_ROOT/$O/pmc_summary.txt | head -12
rm -rf $GRAFT_REPO_ROOT/$O/pmc
