#!/bin/bash
# round 6, call 2: the second version of the wavefront-mapped 64k fast encoder against the first, same box; GPU encode tests on the product build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "encode or limited or fast or stream or frame or wrap or codec" 2>&1 | tail -5 | tee $O/gpu_tests_encode.txt
cp lz4net_amd/liblz4hip.so /tmp/product.so
LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD python lz4net_amd/build.py > $O/build_tuning.log 2>&1 || tail -5 $O/build_tuning.log
LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD timeout 900 python tools/enc_wave_versions_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/encoder_wave_versions_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
