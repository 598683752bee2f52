#!/bin/bash
# round 6, call 12: mid-size batches (2^14 .. 2^17 blocks) leave most of the LDS idle: does the lane decoder with a LARGER ring (fewer far fetches) run them faster?
# tuning build, knob decoder_ring, one block per lane (decoder_persist 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call12; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD python lz4net_amd/build.py > $O/build_tuning.log 2>&1 || tail -5 $O/build_tuning.log
for n in 16384 32768 65536 131072; do
  LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD LZ4HIP_DECODER_PERSIST=2 STEPS=5 timeout 600 python tools/ab_decoder_knobs.py $n "4:59192,4:59256,4:59384,4:59512,4:59768,4:59960,4:59192" "2,3" 2>&1 | grep -v amdgpu.ids
done | tee $O/decoder_mid_batches_large_rings.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
