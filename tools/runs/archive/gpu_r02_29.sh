#!/bin/bash
# round-2 run 29: G2 general addition + doubling of the tail kernels with static bounds (g2x28::add_mem, g2x28::dbl): parity + A/B against the
# same sources built with -DBZK_G2_FAST_TAILS=0 (bazuka_amd/libbzk_ab.so), same box, alternating
set -x
O=gpurun_out/r02_29
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py tests/test_gpu_mpn_prove.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not tree_2p24 and not ntt_2p24" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
timeout 200 python tests/tools/fuzz_gpu.py 40 5 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt | cut -c1-400
cp bazuka_amd/libbzk.so /tmp/libbzk_new.so
leg() {
  timeout 200 python tools/sweep.py child g2 20 2>/dev/null | tail -1 | cut -c1-600 | sed "s/^/$1 /"
}
for rep in 1 2; do
  cp /tmp/libbzk_new.so bazuka_amd/libbzk.so; leg static_bounds
  cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; leg generic
done | tee $O/ab.txt
for which in new ab; do
  if [ $which = new ]; then cp /tmp/libbzk_new.so bazuka_amd/libbzk.so; else cp bazuka_amd/libbzk_ab.so bazuka_amd/libbzk.so; fi
  for s in 1 4; do echo -n "$which "; timeout 200 python tools/pipe_probe.py $s 16 2>/dev/null | tail -1; done
done | tee $O/pipe_probe.txt
cp /tmp/libbzk_new.so bazuka_amd/libbzk.so
tail -3 $O/pytest.txt
echo finished
