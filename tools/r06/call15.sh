#!/bin/bash
# round 6, call 15: workgroups of four wavefronts WITH dual ring stores (the four rings interleaved across 256 lanes): the headline launch, the mid sizes, the decoder tests and fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call15; rm -rf $O; mkdir -p $O
bash tools/r06/call14.sh > /dev/null 2>&1; cp gpurun_out/r06_call14/decoder_headline_workgroup_shape.txt $O/; cat $O/decoder_headline_workgroup_shape.txt
for w in 1 0; do echo "== decoder_wg4 = $w (1: workgroups of one wavefront; 0: of four from more than one wavefront per CU on)"; LZ4HIP_DECODER_WG4=$w timeout 600 python tools/dec_default_vs_forced.py 2 2>&1 | grep "dist"; done | tee $O/decoder_mid_batches_placement_dual_stores.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "decode or decoder or fuzz or unknown or stream or frame or lane or persist or corpus" 2>&1 | tail -4 | tee $O/gpu_tests_decode.txt
timeout 900 python tools/fuzz_gpu_decoders.py 100 400 2>&1 | tail -2 | tee $O/fuzz_gpu_decoders.txt
