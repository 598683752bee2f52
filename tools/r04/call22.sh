#!/bin/bash
# random dependent read-modify-writes over 8 GiB taken as chunks spread over 16 .. 128 GiB of one allocation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call22; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/rs2 tools/microbench_random_sectors_spread.hip || exit 1
for p in 1 2; do echo "== process $p"; timeout 300 /tmp/rs2 3000; done 2>&1 | tee $O/random_sectors_spread.txt
