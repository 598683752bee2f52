#!/bin/bash
# one fresh box per call: does the slab probe predict the lane encoder's rate?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_slab
timeout 600 python tools/enc_slab_probe_vs_rate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_slab/box_$(date +%s).txt
