#!/bin/bash
# PCIe-inclusive host-pointer batch rate vs number of host gather/scatter threads and batch size.
cd $GRAFT_REPO_ROOT
for m in 4096 16384; do
for t in 8 16 32; do
  echo "== blocks $m host threads $t"
  LZ4HIP_HOST_THREADS=$t python tools/host_batch_rate.py $m 2>&1 | grep "^dist"
done
done
