#!/bin/bash
# round 6: host-pointer batches with a quarter slice at both ends (host_taper 0 = default: tapered; 1 = equal slices), one and two pipelines on the device
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_host; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "host or stream or wrap or frame or lz4codec or sharding" 2>&1 | tail -3 > $O/host_tests.txt; cat $O/host_tests.txt
for rep in 1 2; do for w in 1 2; do for t in 1 0; do
  echo "== host_workers $w host_taper $t (repetition $rep)"
  LZ4HIP_HOST_WORKERS=$w LZ4HIP_HOST_TAPER=$t python tools/host_slices_knob_sweep.py 4096,8192,16384,32768,65536 0 2 2>&1 | grep -v amdgpu
done; done; done > $O/host_tapered_slices.txt
cat $O/host_tapered_slices.txt
