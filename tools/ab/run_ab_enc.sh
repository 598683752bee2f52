#!/bin/bash
# Same-box A/B of library variants for the fast encoder: tools/ab/run_ab_enc.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  cp tools/ab/$v lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  python tools/enc_rate.py 262144 2>/dev/null | grep dist | sed "s/^/$v /"
done
done
