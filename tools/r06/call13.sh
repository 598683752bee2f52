#!/bin/bash
# round 6, call 13: the lane decoder's workgroups of four wavefronts for batches of at most eight wavefronts per CU (knob decoder_wg4: 0 on, 1 off): default dispatch and forced
# mappings at 8 192 .. 262 144 blocks, D2 and D3; then the decoder GPU tests and a fuzz run (forced-lane tests take the new form for every batch of >= 193 blocks)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call13; rm -rf $O; mkdir -p $O
for w in 1 0 1 0; do echo "== decoder_wg4 = $w (1: workgroups of one wavefront, as before; 0: of four)"; LZ4HIP_DECODER_WG4=$w timeout 600 python tools/dec_default_vs_forced.py 2 2>&1 | grep "dist"; done | tee $O/decoder_mid_batches_placement.txt
for w in 1 0; do echo "== decoder_wg4 = $w"; LZ4HIP_DECODER_WG4=$w timeout 600 python tools/dec_default_vs_forced.py 3 2>&1 | grep "dist"; done | tee -a $O/decoder_mid_batches_placement.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "decode or decoder or fuzz or unknown or stream or frame or lane or persist or corpus" 2>&1 | tail -4 | tee $O/gpu_tests_decode.txt
timeout 900 python tools/fuzz_gpu_decoders.py 100 400 2>&1 | tail -2 | tee $O/fuzz_gpu_decoders.txt
