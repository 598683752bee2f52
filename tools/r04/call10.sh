#!/bin/bash
cd $GRAFT_REPO_ROOT
export LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD
O=gpurun_out/r04_call10; mkdir -p $O
STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "${CF:-4:11192,4:27192,4:25192}" "2,3" 2>&1 | grep -v amdgpu.ids | tee $O/gen4_alt.txt
STEPS=5 timeout 300 python tools/ab_decoder_knobs.py 262144 "${CF:-4:11192,4:27192,4:25192}" "2,3" 2>&1 | grep -v amdgpu.ids | tee -a $O/gen4_alt.txt
