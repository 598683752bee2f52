#!/bin/bash
# the encoder's two rates: random dependent read-modify-writes over footprints of 0.25 .. 32 GiB (address translation?), three fresh processes,
# and the encoder's own rate in a fresh process on the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call21; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/rs tools/microbench_random_sectors.hip || exit 1
for p in 1 2; do echo "== process $p"; timeout 300 /tmp/rs 3000 placement alignment; done 2>&1 | tee $O/random_sectors.txt
timeout 300 python tools/enc_stream_probe.py 1048576 2 2>&1 | grep -v amdgpu.ids | head -3 | tee $O/encoder_same_box.txt
