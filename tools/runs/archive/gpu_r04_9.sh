#!/bin/bash
# round-4 run 9: Poseidon kernels with the MDS rows as straight-line code (no scratch): parity, same-box A/B against the rolled-loop build, PMC
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_tree4.py tests/test_gpu_mpn_tree.py tests/test_gpu_state_compress.py tests/test_gpu_mpn_devtree.py tests/test_golden_gpu.py -m gpu -q -x > $O/pytest_pos.txt 2>&1; echo "rc=$?" >> $O/pytest_pos.txt
timeout 120 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k tree > $O/pytest_tree24.txt 2>&1; echo "rc=$?" >> $O/pytest_tree24.txt
timeout 600 python tools/sweep.py r4pos > $O/sweep_pos.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_kernel<bzk::G2Fast:gather" > $O/pmc_other.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
tail -4 $O/pytest_pos.txt; tail -3 $O/pytest_tree24.txt; cat $O/sweep_pos.txt | cut -c1-200; cut -c1-1200 $O/pmc_other.log
echo finished
