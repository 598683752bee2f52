#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call9; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 | tee $O/gpu_tests.txt
for m in 4096 16384; do timeout 300 python tools/host_batch_rate.py $m 2>&1 | grep -v amdgpu.ids; done | tee $O/host_rate.txt
