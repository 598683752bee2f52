#!/bin/bash
# the long decoder fuzz (300 seeds x 450 streams x known/unknown x three decoder forms = 810 000 comparisons) on the round-5 lane decoder
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_fuzz
timeout 1500 python tools/fuzz_gpu_decoders.py 300 400 2>&1 | tail -4 | tee gpurun_out/r05_fuzz/fuzz_gpu_decoders_810000_comparisons.txt
