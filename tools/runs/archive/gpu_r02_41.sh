#!/bin/bash
# round-2 run 41: is it the producers' mmap / munmap traffic (glibc serves blocks above 128 KB with mmap and returns them with munmap; every
# munmap runs the GPU driver's MMU notifier) that slows a prover beside them?  same probe with the malloc thresholds raised
O=gpurun_out/r02_41
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
{
echo -n "default malloc, 1 slot + 8x8 producers: "; timeout 90 python tools/pipe_probe.py 1 24 8 8 2>/dev/null | tail -1
echo -n "mmap threshold 32 MB, no trim, 1 slot + 8x8 producers: "; MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=68719476736 MALLOC_TOP_PAD_=268435456 timeout 90 python tools/pipe_probe.py 1 24 8 8 2>/dev/null | tail -1
echo -n "same, 4 slots + 8x8 producers: "; MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=68719476736 MALLOC_TOP_PAD_=268435456 timeout 90 python tools/pipe_probe.py 4 24 8 8 2>/dev/null | tail -1
echo -n "default malloc, 4 slots + 8x8 producers: "; timeout 90 python tools/pipe_probe.py 4 24 8 8 2>/dev/null | tail -1
} > $O/out.txt 2>&1
cat $O/out.txt
echo finished
