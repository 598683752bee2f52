#!/bin/bash
# round 6, call 10: long differential fuzz of the encoders (new) and of the decoders (round 5's 810 000-comparison run) on the final kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call10; rm -rf $O; mkdir -p $O
timeout 1500 python tools/fuzz_gpu_encoders.py 8 606 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/fuzz_gpu_encoders.txt
timeout 1500 python tools/fuzz_gpu_decoders.py 300 400 2>&1 | tail -3 | tee $O/fuzz_gpu_decoders_810000_comparisons.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "second_version" 2>&1 | tail -3 | tee $O/gpu_test_encoder_paths.txt
