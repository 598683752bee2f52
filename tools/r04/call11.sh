#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call11; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or roundtrip or stream or fuzz or arbitrary or unaligned or canar or error" 2>&1 | tail -5 | tee $O/gpu_tests_decode.txt
timeout 600 python tools/decode_threshold.py "2,3,0,1" "1024,4096,16384,65536" 2>&1 | grep -v amdgpu.ids | tee $O/decode_small_batches.txt
for m in 4096 16384; do timeout 300 python tools/host_batch_rate.py $m 2>&1 | grep -v amdgpu.ids; done | tee $O/host_rate.txt
