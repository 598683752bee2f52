#!/bin/bash
# PCIe-inclusive host-pointer batch rate vs number of slices the batch is cut into (kernels of neighbouring slices overlap).
cd $GRAFT_REPO_ROOT
for m in 1024 4096 16384; do
for sl in 3 6 8 12; do
  echo "== blocks $m slices $sl"
  LZ4HIP_HOST_SLICES=$sl python tools/host_batch_rate.py $m 2>&1 | grep "^dist"
done
done
