#!/bin/bash
# Quick A/B of the working tree's library on the headline workload: D2 and D3 decode at 2^20 blocks, 5 steps each.
cd $GRAFT_REPO_ROOT
for d in ${DISTS:-2 3}; do
  r=$(python bench.py --no-cpu --no-extras --hc-blocks 0 --steps 5 --dist $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['mean_kernel_ms'], d['verified'])")
  echo "dist=$d: $r"
done
