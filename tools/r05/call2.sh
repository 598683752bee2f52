#!/bin/bash
# round 5, call 2: sector input + dual ring stores with larger rings (fewer wavefronts per CU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call2; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
for v in tune_dual2; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"
  STEPS=3 timeout 900 python tools/ab_decoder_knobs.py 1048576 "4:59192,4:59208,4:59224,4:59256,4:59192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/ab_rings.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
