#!/bin/bash
# round-6 run 31: NTT: the second inter-pass twiddle of a three-pass plan from a single-level table (one load instead of lo * hi): parity, then A/B (BZK_NTT_NO_TONE=1)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run31; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_poseidon_ntt.py tests/test_gpu_fullsize.py tests/test_gpu_groth16.py tests/test_gpu_production.py -m gpu -q --timeout=420 --durations=4 -x -k "not 1024tx" ) > $O/pytest_ntt.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ntt.txt
tail -4 $O/pytest_ntt.txt | cut -c1-200
timeout 100 python tests/tools/fuzz_gpu.py 30 3131 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-300
for rep in 1 2; do for off in 1 0; do
for lg in 21 22 23 24 25 26; do BZK_NTT_NO_TONE=$off timeout 200 python tools/sweep.py child ntt $lg | grep '^{' | sed "s/^{/{\"no_tone\": $off, /"; done
for lg in 21 22 24; do BZK_NTT_NO_TONE=$off timeout 200 python tools/sweep.py child h $lg | grep '^{' | sed "s/^{/{\"no_tone\": $off, /"; done
done; done > $O/ntt_ab.txt 2>&1
cut -c1-200 $O/ntt_ab.txt
echo finished
