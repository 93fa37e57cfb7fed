#!/bin/bash
# round-4 run 14: soak (800 proofs through 4 slots, every proof verified) and a second fuzz seed on the committed build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_run14
timeout 400 python tests/tools/soak.py 800 4 > gpurun_out/r04_run14/soak.txt 2>&1; echo "soak rc=$?" >> gpurun_out/r04_run14/soak.txt
timeout 300 python tests/tools/fuzz_gpu.py 100 77 > gpurun_out/r04_run14/fuzz.txt 2>&1; echo "fuzz rc=$?" >> gpurun_out/r04_run14/fuzz.txt
tail -3 gpurun_out/r04_run14/soak.txt | cut -c1-900; tail -2 gpurun_out/r04_run14/fuzz.txt | cut -c1-500
