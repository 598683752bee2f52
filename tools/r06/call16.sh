#!/bin/bash
# round 6, call 16: wavefront-mapped fast encoder with four blocks per workgroup (knob encoder_wg4) against one, small and mid batches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call16; rm -rf $O; mkdir -p $O
for w in 0 1 0 1; do echo "== encoder_wg4 = $w"; LZ4HIP_ENCODER_WG4=$w timeout 600 python tools/enc_wave_rates.py 2>&1 | grep -v amdgpu.ids | grep "wavefront encoder"; done | tee $O/wave_encoder_workgroup_shape.txt
