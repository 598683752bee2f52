#!/bin/bash
# what bounds the lane decoder's iteration rate: N more vector-ALU instructions / N more LDS stores per iteration (prebuilt diagnostic libraries)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call16; mkdir -p $O
for v in base valu20 valu40 lds7 lds14 base; do
  cp build_variants/ballast_$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"
  STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "4:27192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/ballast.txt
