#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call4; mkdir -p $O
CF="3:128,4:128,4:2128,4:192,4:2192,4:3192,4:1192,4:256,4:1256,4:3256,4:1240"
STEPS=3 timeout 600 python tools/ab_decoder_knobs.py 1048576 "$CF" "2,3" > $O/gen4_2p20.txt 2>&1
cat $O/gen4_2p20.txt
STEPS=3 timeout 300 python tools/ab_decoder_knobs.py 262144 "3:128,4:192,4:1192,4:1256,4:3192" "2,3" > $O/gen4_2p18.txt 2>&1
cat $O/gen4_2p18.txt
