#!/bin/bash
# (build_variants/host_shaped.so / host_uniform.so: the library with tools/ab/host_pipeline_slot_streams_shaped_slices.patch applied,
#  the second one with -DLZ4HIP_HOST_UNIFORM_SLICES=1)
# host-pointer pipeline: one stream per slot, shaped slices (small first and last slices) against equal slices, same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call20; mkdir -p $O
for v in uniform shaped uniform shaped; do
  cp build_variants/host_$v.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
  echo "== $v"
  for m in 16384 32768; do timeout 300 python tools/host_batch_rate.py $m 2>&1 | grep -v amdgpu.ids; done
done 2>&1 | tee $O/host_shaped_vs_uniform.txt
cp build_variants/host_shaped.so lz4net_amd/liblz4hip.so; touch lz4net_amd/liblz4hip.so
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python tools/host_decode_timeline.py 16384 2 2>&1 | grep "decode call\|ok=" | tee $O/traced.txt
python tools/host_timeline_report.py $O/trace | tee $O/host_decode_timeline_D2_16384.txt | tail -50
rm -rf $O/trace
