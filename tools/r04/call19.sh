#!/bin/bash
# device-side timeline of a 16 384-block host-pointer decode (D2)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call19; mkdir -p $O
timeout 300 python tools/host_decode_timeline.py 16384 2 2>&1 | grep -v amdgpu.ids | tee $O/untraced.txt
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python tools/host_decode_timeline.py 16384 2 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/traced.txt
python tools/host_timeline_report.py $O/trace | tee $O/host_decode_timeline_D2_16384.txt | tail -60
find $O/trace -name "*.csv" -size +2M -delete; find $O/trace -name "*.db" -delete
