#!/bin/bash
# round 6, call 7: wavefront encoder with the search's first-step words and the literals requested ahead (b) against the second version without them (a), same box;
# the encoder GPU tests on the product library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call7; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
export LZ4HIP_KEEP_LIBRARY=1
for v in enc_wave_a enc_wave_b enc_wave_a enc_wave_b; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so
  echo "== $v"; timeout 600 python tools/enc_wave_rates.py 2>&1 | grep -v amdgpu.ids
done | tee $O/wave_encoder_prefetch_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
unset LZ4HIP_KEEP_LIBRARY
timeout 1200 python -m pytest tests -m gpu -x -q -k "encode or limited or fast or stream or frame or wrap or codec or corpus" 2>&1 | tail -4 | tee $O/gpu_tests_encode.txt
