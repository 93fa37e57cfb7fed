#!/bin/bash
# round-6 run 46: the issue budget of a 16-tx proof on the SHIPPED build (one --pmc pass over tools/prove_serial.py -> tools/valu_budget.py), against the pipelined time per proof
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_run46; mkdir -p $O
export TMPDIR=/tmp
BZK_PROVE_SERIAL=1 timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_serial -- python tools/prove_serial.py 4 > $O/pmc_serial.log 2>&1
T=$(find $O/pmc_serial -name "*.db" | head -1); python tools/valu_budget.py $T 4 12.9 > $O/valu_budget.txt 2>&1
cut -c1-170 $O/valu_budget.txt | head -45
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
AMD_DIRECT_DISPATCH=0 BZK_SYNC_BLOCKING=1 timeout 300 python tools/pipe_probe.py 4 24 2>/dev/null | tail -1
echo finished
