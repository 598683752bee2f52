#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_wave_decoder; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
for v in wave_old wave_new wave_old wave_new; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"; timeout 600 python tools/wave_decode_rates.py 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
