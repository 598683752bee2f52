#!/bin/bash
# can separate allocations with temporary spacers get the spread placement?  empty device, and with 192 GiB held by the application
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call23; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/rs3 tools/microbench_random_sectors_spacers.hip || exit 1
{ timeout 300 /tmp/rs3 3000 0; timeout 300 /tmp/rs3 3000 0; timeout 300 /tmp/rs3 3000 192; } 2>&1 | tee $O/random_sectors_spacers.txt
