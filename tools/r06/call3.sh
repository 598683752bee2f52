#!/bin/bash
# round 6, call 3: where a lone wavefront of the wavefront-mapped fast encoder spends its cycles (s_memtime sections), D2 and D3, 512 and 2560 blocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call3; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/enc_wave_sections.hip -o /tmp/enc_wave_sections 2>/dev/null
for d in 2 3; do for n in 512 2560; do timeout 120 /tmp/enc_wave_sections $n $d; done; done 2>&1 | tee $O/encoder_wave_sections.txt
