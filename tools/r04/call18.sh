#!/bin/bash
# the fast encoder's two rates: same process, same slab, different streams; three fresh processes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call18; mkdir -p $O
for p in 1 2 3; do echo "== process $p"; timeout 300 python tools/enc_stream_probe.py 1048576 5 2>&1 | grep -v amdgpu.ids; done | tee $O/encoder_streams.txt
