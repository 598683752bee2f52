#!/bin/bash
# the memory skeleton of the lane decoder (tools/decode_skeleton.hip, round 4) re-run for the SHIPPED access pattern: whole sectors of input, ring 192,
# 128-byte flush units, twelve wavefronts per CU -- without arithmetic and with the real kernel's ~290 vector-ALU + ~20 LDS instructions per iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_skeleton
./tools/decode_skeleton 20 "2:1:192:12:0:0:128,2:3:192:12:0:0:128,2:3:192:12:150:10:128,2:3:192:12:290:20:128,2:3:192:12:250:20:128,3:1:192:12:0:0:128,3:3:192:12:0:0:128,3:3:192:12:290:20:128" 2>&1 | tee gpurun_out/r05_skeleton/decode_skeleton_shipped_configuration_2p20.txt
