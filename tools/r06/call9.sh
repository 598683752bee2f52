#!/bin/bash
# round 6, call 9: width of a search's first step in the wavefront encoder: 8 / 16 (b) / 32 / 64 probes, same box, against the second version as of 84019f8 (a)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call9; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
export LZ4HIP_KEEP_LIBRARY=1
for v in enc_wave_w64 enc_wave_b enc_wave_w64 enc_wave_b; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so
  echo "== $v"; timeout 600 python tools/enc_wave_rates.py 2>&1 | grep -v amdgpu.ids
done | tee $O/wave_encoder_adaptive_first_step_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
