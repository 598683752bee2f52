#!/bin/bash
# round 6: host-pointer fast encode as ONE pipeline of equal slices of at most one residency round (default) against two pipelines (host_workers 2) and the old equal slices (host_slices 6 / 4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_host; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "host or stream or wrap or frame or lz4codec or sharding" 2>&1 | tail -3 > $O/host_tests2.txt; cat $O/host_tests2.txt
for rep in 1 2 3; do
  echo "== default (repetition $rep)"
  python tools/host_slices_knob_sweep.py 4096,8192,12288,16384,32768 0 2 2>&1 | grep -v amdgpu
  echo "== host_workers 2 (repetition $rep)"
  LZ4HIP_HOST_WORKERS=2 python tools/host_slices_knob_sweep.py 4096,8192,12288,16384,32768 0 2 2>&1 | grep -v amdgpu
  echo "== host_workers 1, host_slices 6 = round 5's rule from 12 288 blocks up (repetition $rep)"
  LZ4HIP_HOST_WORKERS=1 python tools/host_slices_knob_sweep.py 12288,16384,32768 6 2 2>&1 | grep -v amdgpu
done > $O/encode_one_pipeline.txt
cat $O/encode_one_pipeline.txt
