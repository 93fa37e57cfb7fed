#!/bin/bash
# round-4 run 5: diagnosis of the endomorphism form's slower accumulation: task length / reduction chunk sweeps, L2 hit rates and stall counters
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/sweep.py r4endo2 > $O/sweep_endo2.txt 2>&1
for e in 0 1; do
  BZK_MSM_ENDO_G2=$e timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/pmc_tcc_$e -- python tools/sweep.py child g2res 20 > $O/pmc_tcc_$e.log 2>&1
  BZK_MSM_ENDO_G2=$e timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_sq_$e -- python tools/sweep.py child g2res 20 > $O/pmc_sq_$e.log 2>&1
  for k in tcc sq; do T=$(find $O/pmc_${k}_$e -name "*.db" | head -1); python tools/rocpd_summary.py $T > $O/pmc_${k}_$e.txt 2>&1; done
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cut -c1-700 $O/sweep_endo2.txt
for e in 0 1; do for k in tcc sq; do echo "== $k endo=$e"; grep -i "accumulate" $O/pmc_${k}_$e.txt | head -12; done; done
echo finished
