#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_hc_sweep
timeout 1500 python tools/hc_knob_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_hc_sweep/hc_knob_sweep.txt
