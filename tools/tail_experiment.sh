#!/bin/bash
# Is the lane decoder's launch paying for a partially filled last round of wavefronts?  2^20 blocks = 16384 wavefronts on
# 3072 resident slots (12 per CU x 256) = 5.33 rounds; 983040 blocks = exactly 5 rounds, 1179648 = exactly 6.
cd $GRAFT_REPO_ROOT
for d in 2 3; do
for n in 983040 1048576 1179648; do
  r=$(python bench.py --no-cpu --no-extras --hc-blocks 0 --steps 5 --dist $d --blocks $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['mean_kernel_ms'], d['verified'])")
  echo "dist=$d blocks=$n: $r"
done
done
