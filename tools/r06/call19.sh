#!/bin/bash
# round 6: host-pointer fast encode with slices of one residency round (host_slices 0 = the new automatic rule; 4 / 6 = equal slices, the old rule's values)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_host; mkdir -p $O
for rep in 1 2; do for w in 1 2; do
  echo "== host_workers $w (repetition $rep)"
  LZ4HIP_HOST_WORKERS=$w python tools/host_slices_knob_sweep.py 4096,8192,16384,32768 0,4,6 2 2>&1 | grep -v amdgpu
done; done > $O/encode_round_slices.txt
cat $O/encode_round_slices.txt
