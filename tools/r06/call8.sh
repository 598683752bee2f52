#!/bin/bash
# round 6, call 8: wavefront encoder, second version as of 84019f8 (a) against the working tree (b: loads issued unconditionally from clamped addresses,
# the window re-based before them, first-step words and literals requested ahead); section timers of (b)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call8; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
export LZ4HIP_KEEP_LIBRARY=1
for v in enc_wave_a enc_wave_b enc_wave_a enc_wave_b; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so
  echo "== $v"; timeout 600 python tools/enc_wave_rates.py 2>&1 | grep -v amdgpu.ids
done | tee $O/wave_encoder_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
unset LZ4HIP_KEEP_LIBRARY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/enc_wave_sections.hip -o /tmp/enc_wave_sections 2>/dev/null
for d in 2 3; do for n in 512 2560; do timeout 120 /tmp/enc_wave_sections $n $d; done; done 2>&1 | tee $O/encoder_wave_sections.txt
