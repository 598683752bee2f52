#!/bin/bash
# round 5, call 5: LZ4HC A/B on one box: table entries without / with the mismatch byte
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call5; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
for v in hc_old hc_new hc_old hc_new; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"
  timeout 900 python tools/hc_rate.py 262144 16,16 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/hc_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
