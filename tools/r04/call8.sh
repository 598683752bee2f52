#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call8; mkdir -p $O
timeout 600 python tools/hc_sub_chunks_ab.py 262144 "1,2,4,8" 2>&1 | grep -v amdgpu.ids | tee $O/hc_sub_chunks.txt
for m in 4096 16384; do timeout 300 python tools/host_batch_rate.py $m 2>&1 | grep -v amdgpu.ids; done | tee $O/host_rate.txt
for t in 16 32 64 128; do echo "host_threads=$t"; LZ4HIP_HOST_THREADS=$t timeout 300 python tools/host_batch_rate.py 16384 2>&1 | grep "dist 2"; done | tee -a $O/host_rate.txt
