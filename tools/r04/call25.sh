#!/bin/bash
# random dependent READS (the LZ4HC lane kernel's pattern: 256 KiB tables, 64 GiB in all) contiguous against spread; two processes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call25; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/rs4 tools/microbench_random_sectors_reads.hip || exit 1
for p in 1 2; do echo "== process $p"; timeout 300 /tmp/rs4 3000; done 2>&1 | tee $O/random_sectors_reads.txt
