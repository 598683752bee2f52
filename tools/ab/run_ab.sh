#!/bin/bash
# Same-box A/B of library variants: tools/ab/run_ab.sh libA.so libB.so ...   (prints D2 and D3 decode GB/s per variant)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  cp tools/ab/$v lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  for d in 2 3; do
    r=$(python bench.py --no-cpu --no-extras --hc-blocks 0 --steps 5 --dist $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['verified'])")
    echo "$v dist=$d: $r"
  done
done
done
