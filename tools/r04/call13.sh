#!/bin/bash
# compiler scheduling strategies for the whole library, A/B on the lane decoder (built on the GPU box; nothing is kept)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call13; mkdir -p $O
for f in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-metric-bias=0" ; do
  echo "== flags: [$f]"
  LZ4HIP_BUILD_FLAGS="$f" python -c "from lz4net_amd import build; build.build(force=True)" 2>&1 | grep -E " error|rror:" | head -3
  STEPS=3 timeout 300 python tools/ab_decoder_knobs.py 1048576 "4:27192" "2,3" 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/sched_flags.txt
