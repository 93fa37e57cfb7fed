#!/bin/bash
# run 47: BASELINE configs[2] at face value - ONE UpdateCircuit of 1024 transactions (L=15, T=3, B=5): 57.8 M constraints,
# NTT domain 2^26, CRS generated on the GPU (~33 GB), proof on one MI355X, pairing-checked by the oracle
set -x
mkdir -p gpurun_out/r47
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
nproc > gpurun_out/r47/host.txt; free -g >> gpurun_out/r47/host.txt; rocm-smi --showmeminfo vram >> gpurun_out/r47/host.txt 2>&1
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo)
echo "MemAvailable ${avail} GiB" >> gpurun_out/r47/host.txt
if [ "$avail" -lt 300 ]; then echo "not enough host memory for the 2^26 instance: skipped" >> gpurun_out/r47/host.txt; exit 0; fi
BZK_DEBUG=1 timeout 1000 python tests/tools/prove_production.py 5 2 0 > gpurun_out/r47/production_1024tx.txt 2> gpurun_out/r47/production_1024tx_err.txt
echo "rc=$?" >> gpurun_out/r47/production_1024tx.txt
tail -c 3000 gpurun_out/r47/production_1024tx.txt; tail -c 1500 gpurun_out/r47/production_1024tx_err.txt
echo finished
