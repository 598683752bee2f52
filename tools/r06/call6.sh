#!/bin/bash
# round 6, call 6: wavefront decoder with the wider burst eligibility (<= 14 literals, one match-length byte) against the previous one, same box; the decoder tests and a
# slice of the long fuzz on the new one; host-pointer rates with two staging pipelines per device
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call6; rm -rf $O; mkdir -p $O
cp lz4net_amd/liblz4hip.so /tmp/product.so
export LZ4HIP_KEEP_LIBRARY=1
for v in wave_dec_old wave_dec_new wave_dec_old wave_dec_new; do
  cp build_variants/$v.so lz4net_amd/liblz4hip.so
  echo "== $v"; timeout 600 python tools/wave_decode_rates.py 2>&1 | grep -v amdgpu.ids
done | tee $O/wave_decoder_burst_eligibility_ab.txt
cp /tmp/product.so lz4net_amd/liblz4hip.so
unset LZ4HIP_KEEP_LIBRARY
timeout 1200 python -m pytest tests -m gpu -x -q -k "decode or decoder or fuzz or stream or frame or unknown or host or multi" 2>&1 | tail -6 | tee $O/gpu_tests_decode.txt
timeout 900 python tools/fuzz_gpu_decoders.py 40 400 2>&1 | tail -3 | tee $O/fuzz_gpu_decoders.txt
for w in 1 2 3 4; do echo "== host_workers $w"; LZ4HIP_HOST_WORKERS=$w timeout 300 python tools/wave_decode_rates.py 2>&1 | grep host-pointer; done | tee $O/host_workers_sweep.txt
