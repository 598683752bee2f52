#!/bin/bash
# round 5, call 1: the new GPU tests (counter slots across streams, fuzz slice) + same-box A/B of the sector-input lane decoder
# (decoder_ring bit 5) with and without the dual ring stores (build_variants/*.so, tuning builds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_device_batch.py tests/test_gpu_parity.py -m gpu -x -q -s -k "counter_slots or fuzz_slice or persistent_many or release_workspaces" 2>&1 | tail -8 | tee $O/new_tests.txt
cp lz4net_amd/liblz4hip.so /tmp/product.so
for v in tune_base tune_dual2; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"
  STEPS=3 timeout 900 python tools/ab_decoder_knobs.py 1048576 "4:27192,4:59192,4:43192,4:35192,4:58128,4:27192,4:59192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/ab_sector_input.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
