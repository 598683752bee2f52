#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call5; mkdir -p $O
CF="${CF:-3:128,4:3192,4:7192,4:1192,4:5192,4:2192,4:3256,4:7256,4:128}"
STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "$CF" "2,3" > $O/gen4_2p20.txt 2>&1
cat $O/gen4_2p20.txt
