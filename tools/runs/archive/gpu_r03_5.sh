set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_run5; mkdir -p $O
for v in 2 4 5; do BZK_NTT_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_groth16.py -x -q -m gpu -k "ntt or h_chain or h_stage" > $O/pytest_ntt_v$v.log 2>&1; echo "variant $v rc=$?" >> $O/pytest_ntt_v$v.log; tail -2 $O/pytest_ntt_v$v.log; done
timeout 1200 python tools/sweep.py r3ntt2 > $O/sweep_ntt2.log 2>&1
cat $O/sweep_ntt2.log
cd /tmp && export TMPDIR=/tmp
for v in 2 4; do
BZK_NTT_VARIANT=$v timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_v$v -- python $GRAFT_REPO_ROOT/tools/sweep.py child ntt 20 > $GRAFT_REPO_ROOT/$O/pmc_v$v.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$O/pmc_v$v -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/pmc_v${v}_summary.txt 2>&1
grep -i "ntt_" $GRAFT_REPO_ROOT/$O/pmc_v${v}_summary.txt | head -30
done
rm -rf $GRAFT_REPO_ROOT/$O/pmc_v2 $GRAFT_REPO_ROOT/$O/pmc_v4
