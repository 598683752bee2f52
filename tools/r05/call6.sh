#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call6; rm -rf $O; mkdir -p $O
for n in 16384 4096 2731; do timeout 600 python tools/enc_mid_batch_split.py $n 2 2>&1 | grep -v amdgpu.ids; done | tee $O/enc_mid_split.txt
timeout 600 python tools/enc_mid_batch_split.py 16384 3 2>&1 | grep -v amdgpu.ids | tee -a $O/enc_mid_split.txt
