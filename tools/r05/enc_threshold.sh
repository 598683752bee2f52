#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_enc_threshold
timeout 900 python tools/enc_threshold.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_enc_threshold/enc_threshold.txt
