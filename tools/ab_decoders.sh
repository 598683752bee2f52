#!/bin/bash
# Same-box A/B of decoder configurations on the full-size bench workload.
# usage: bash tools/ab_decoders.sh <tag> "<decoder>:<ring>[:ENV=VAL,...]" ...      -> gpurun_out/<tag>.txt
tag=$1; shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag.txt
: > $out
for cfg in "$@"; do
  IFS=: read dec ring extra <<< "$cfg"
  for d in ${DISTS:-2 3}; do
    envs="LZ4HIP_RING_BYTES=$ring"
    [ -n "$extra" ] && envs="$envs ${extra//,/ }"
    r=$(env $envs python bench.py --no-cpu --no-extras --hc-blocks 0 --steps ${STEPS:-5} --warmup 1 --dist $d --decoder $dec 2>>gpurun_out/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['verified'])")
    echo "$cfg dist=$d: $r" | tee -a $out
  done
done
