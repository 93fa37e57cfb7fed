#!/bin/bash
# round-4 run 27: instruction-cache behaviour of the prover's kernels - alone (BZK_PROVE_SERIAL=1, one kernel at a time) and pipelined
# (four slots: kernels of different proofs share the CUs' instruction caches)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_run27; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BZK_PROVE_SERIAL=1 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $O/serial -- python tools/prove_serial.py 4 > $O/serial.log 2>&1
T=$(find $O/serial -name "*.db" | head -1)
python tools/pmc_generic.py $T SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --ratio SQC_ICACHE_MISSES/SQC_ICACHE_REQ --top 30 > $O/icache_serial.txt 2>&1
BZK_BENCH_TWO_PROCS=0 timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d $O/pipe -- python bench.py --steps 3 --warmup 1 --no-others --no-overlap --no-cpu-baseline --no-production > $O/pipe.log 2>&1
T=$(find $O/pipe -name "*.db" | head -1)
python tools/pmc_generic.py $T SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --ratio SQC_ICACHE_MISSES/SQC_ICACHE_REQ --top 30 > $O/icache_pipelined.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +200k -delete
cut -c1-170 $O/icache_serial.txt | head -24; cut -c1-170 $O/icache_pipelined.txt | head -24; tail -2 $O/pipe.log | cut -c1-300
echo finished
