#!/bin/bash
# fast encoder: process-to-process reproducibility, early vs late workspace allocation, 16 vs 24 wavefronts per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call7; mkdir -p $O
for rep in 1 2 3; do
for early in 1 0; do
for wpc in 24 16; do
  r=$(LZ4HIP_ENCODER_WAVES_PER_CU=$wpc timeout 200 python bench.py --no-cpu --no-extras --hc-blocks 0 --steps 1 --warmup 0 --early-workspace $early 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=list(d['extras'].values())[0]; print(e['encode_fast_GBps'], d['value'])")
  echo "rep=$rep early=$early waves_per_cu=$wpc: encode_fast_GBps decode_value = $r"
done; done; done | tee $O/encoder_repro.txt
