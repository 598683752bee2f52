#!/bin/bash
# round 5, call 3: the new default lane decoder (sector input + dual ring stores): decoder GPU tests, 60-seed fuzz, quick rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_call3; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "decode or decoder or fuzz or persistent or roundtrip or unaligned or smoke or stream or codec" 2>&1 | tail -6 | tee $O/decoder_tests.txt
timeout 900 python tools/fuzz_gpu_decoders.py 60 400 2>&1 | tail -3 | tee $O/fuzz.txt
bash tools/ab_now.sh 2>&1 | tee $O/rates.txt
