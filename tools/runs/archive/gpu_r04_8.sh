#!/bin/bash
# round-4 run 8: NTT with 32-byte inter-pass elements: parity (every NTT / h-chain test, 2^24 vs oracle), A/B timing, PMC traffic of the new build
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run8; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_groth16.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest_ntt.txt 2>&1; echo "rc=$?" >> $O/pytest_ntt.txt
timeout 600 python tools/sweep.py r4ntt > $O/sweep_ntt.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_ops_fetch -- python tools/pmc_ops.py > $O/pmc_ops_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_ops_write -- python tools/pmc_ops.py > $O/pmc_ops_write.log 2>&1
OF=$(find $O/pmc_ops_fetch -name "*.db" | head -1); OW=$(find $O/pmc_ops_write -name "*.db" | head -1)
OSTAMP=$(python -c "import bench; print(bench.other_source_stamp())")
python tools/pmc_kernels.py $OF $OW $O/pmc_other_kernels.json --stamp $OSTAMP --command "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python tools/pmc_ops.py" "ntt_2p24=ntt_pass_kernel:stream:3" "tree_2p24=poseidon29:stream:2" "msm_accumulate_g2=msm_accumulate_kernel<bzk::G2Fast:gather" > $O/pmc_other.log 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
tail -5 $O/pytest_ntt.txt; cat $O/sweep_ntt.txt | cut -c1-300; cut -c1-700 $O/pmc_other.log
echo finished
