#!/bin/bash
# round-6 run 3: (A) the issue budget of a proof (VERDICT r5 item 2 "first MEASURE it"): one PMC pass over tools/prove_serial.py -> tools/valu_budget.py;
# (B) env-only variations of the pipelined ceiling with 16 hardware queues: slots 4 / 5 / 6, BZK_DEDUP_K, BZK_PROVE_LANES=4
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run3; mkdir -p $O
export TMPDIR=/tmp
BZK_PROVE_SERIAL=1 timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_serial -- python tools/prove_serial.py 4 > $O/pmc_serial.log 2>&1
T=$(find $O/pmc_serial -name "*.db" | head -1); python tools/valu_budget.py $T 4 14.1 > $O/valu_budget.txt 2>&1
cat $O/valu_budget.txt | cut -c1-150
export GPU_MAX_HW_QUEUES=16
for S in 4 5 6; do
  echo "== slots $S" >> $O/pipe.txt
  timeout 300 python tools/pipe_probe.py $S 24 >> $O/pipe.txt 2>$O/pipe_err_s$S.txt
done
for K in 16 32; do
  echo "== slots 4 BZK_DEDUP_K=$K" >> $O/pipe.txt
  BZK_DEDUP_K=$K timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_k$K.txt
done
echo "== slots 4 BZK_PROVE_LANES=4" >> $O/pipe.txt
BZK_PROVE_LANES=4 timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_l4.txt
echo "== slots 4 BZK_MSM_ENDO_G1=0 BZK_MSM_ENDO_G2=0 (plain windows)" >> $O/pipe.txt
BZK_MSM_ENDO_G1=0 BZK_MSM_ENDO_G2=0 timeout 300 python tools/pipe_probe.py 4 24 >> $O/pipe.txt 2>$O/pipe_err_noendo.txt
cat $O/pipe.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
echo finished
