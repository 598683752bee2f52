#!/bin/bash
# out-of-range LDS semantics on the hardware, then the lane decoder with dual ring stores against wrapped rows (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call14; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_oor tools/lds_out_of_range.hip && timeout 120 /tmp/lds_oor > $O/lds_out_of_range.txt 2>&1; echo "probe rc=$?" >> $O/lds_out_of_range.txt
cat $O/lds_out_of_range.txt
for f in "-DLZ4HIP_DEC4_DUAL_STORE=0" "-DLZ4HIP_DEC4_DUAL_STORE=1" "-DLZ4HIP_DEC4_DUAL_STORE=0" "-DLZ4HIP_DEC4_DUAL_STORE=1"; do
  echo "== flags: [$f]"
  LZ4HIP_BUILD_FLAGS="$f" python -c "from lz4net_amd import build; build.build(force=True)" 2>&1 | grep -E "rror:" | head -3
  STEPS=3 timeout 300 python tools/ab_decoder_knobs.py 1048576 "4:27192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/dual_store_ab.txt
timeout 600 python -m pytest tests/test_gpu_device_batch.py -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests_dual_store.txt
