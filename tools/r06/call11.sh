#!/bin/bash
# round 6, call 11: the wavefront decoder's s_memtime sections on the current kernel, and with the burst's global store left out (wrong output: what do that store and the
# wait it causes at the top of the next trip cost?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call11; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/dec_wave_sections.hip -o /tmp/dec_a 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc -DLZ4HIP_DEC_EXPERIMENT_NO_BURST_STORE tools/dec_wave_sections.hip -o /tmp/dec_b 2>/dev/null
for n in 1024 4096; do echo "== current kernel"; timeout 120 /tmp/dec_a $n 2; echo "== burst's global store left out"; timeout 120 /tmp/dec_b $n 2; done 2>&1 | tee $O/decoder_wave_sections_store_experiment.txt
