#!/bin/bash
cd $GRAFT_REPO_ROOT
export LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD
O=gpurun_out/r04_call6; mkdir -p $O
for n in 262144 131072 65536; do
STEPS=5 timeout 300 python tools/ab_decoder_knobs.py $n "4:7192,4:3192,4:2128,4:6128,4:128" "2,3" 2>&1 | grep -v amdgpu.ids
done > $O/gen4_mid.txt 2>&1
cat $O/gen4_mid.txt
