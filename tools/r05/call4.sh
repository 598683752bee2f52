#!/bin/bash
# round 5, call 4: LZ4HC with the mismatch byte in the table entries: rate at 2^18 blocks, every block against the CPU reference
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call4; rm -rf $O; mkdir -p $O
timeout 900 python tools/hc_rate.py 262144 16,16 2>&1 | grep -v amdgpu.ids | tee $O/hc_rate.txt
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "full_size_hc or hc_encode_bit_exact or hc_sub_chunks or hc_lane or hc_precomputed or limited_output" 2>&1 | tail -12 | tee $O/hc_tests.txt
