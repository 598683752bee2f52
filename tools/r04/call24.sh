#!/bin/bash
# the lane encoder's slab: measured candidate placements (default: up to 4, stops at the first good one) against the first unmeasured one (1); fresh processes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call24; mkdir -p $O
for t in 1 0 1 0 0; do timeout 300 python tools/enc_slab_calibration.py $t 2>&1 | grep -v amdgpu.ids; done | tee $O/encoder_slab_calibration.txt
