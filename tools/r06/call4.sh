#!/bin/bash
# round 6, call 4: fast encode of large batches with the handed-over blocks shared between the lane grid and a persistent wavefront grid, A/B against the lane grid alone
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call4; rm -rf $O; mkdir -p $O
timeout 1500 python tools/enc_shared_handover_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/encoder_shared_handover_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "encode or limited or fast or corpus" 2>&1 | tail -5 | tee $O/gpu_tests_encode.txt
