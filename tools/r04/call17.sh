#!/bin/bash
# ring stores: wrapped rows (0) / stored twice (1) / stored twice, two rows per LDS instruction (2) x 32- and 64-byte input pieces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call17; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_oor tools/lds_out_of_range.hip && timeout 120 /tmp/lds_oor > $O/lds_out_of_range.txt 2>&1; echo "probe rc=$?" >> $O/lds_out_of_range.txt
cat $O/lds_out_of_range.txt
for d in 0 1 2 0 1 2; do
  cp build_variants/tuning_dual$d.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== dual store = $d"
  STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "4:27192,4:25192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/dual_store_configs.txt
