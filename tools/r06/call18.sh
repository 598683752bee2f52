#!/bin/bash
# round 6: host-pointer encode / decode, workers x slices (each worker cuts ITS shard into `host_slices` slices; 0 = automatic)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_host; mkdir -p $O
for w in 1 2 3; do
  echo "== host_workers $w"
  LZ4HIP_HOST_WORKERS=$w python tools/host_slices_knob_sweep.py 4096,8192,16384,32768 0,1,2,3,4,6,8 2 2>&1 | grep -v amdgpu
done > $O/workers_x_slices.txt
cat $O/workers_x_slices.txt
