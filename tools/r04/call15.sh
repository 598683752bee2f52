#!/bin/bash
# dual ring stores x lane-decoder configurations (prebuilt tuning libraries, same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call15; mkdir -p $O
for d in 0 1 0 1; do
  cp build_variants/tuning_dual$d.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== dual store = $d"
  STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "4:27192,4:25192,4:11192,4:11256" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/dual_store_configs.txt
