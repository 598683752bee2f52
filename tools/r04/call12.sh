#!/bin/bash
cd $GRAFT_REPO_ROOT
rocm-smi --showmemuse --showclocks 2>/dev/null | head -30
for rep in 1 2; do for early in 0 1; do
  r=$(timeout 200 python bench.py --no-cpu --no-extras --hc-blocks 0 --steps 1 --warmup 0 --early-workspace $early 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=list(d['extras'].values())[0]; print(e['encode_fast_GBps'], d['value'])")
  echo "rep=$rep early=$early: encode_fast_GBps decode_value = $r"
done; done
rocm-smi --showmemuse --showclocks --showpower 2>/dev/null | head -40
