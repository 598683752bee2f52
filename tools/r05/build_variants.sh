#!/bin/bash
# The prebuilt library variants tools/r05/call1.sh / call2.sh / call5.sh copy over lz4net_amd/liblz4hip.so (run HERE, in the build container, before
# the gpurun call: hipcc cross-compiles; build_variants/ is git-ignored but travels to the GPU box).
#   tune_base.so   tuning build (every lane-decoder configuration), ring rows wrapped (round 4's appends)
#   tune_dual2.so  tuning build, ring rows stored twice (ds_write2st64_b32), the round-5 default
#   hc_old.so      the library with lz4hip_hc_lcp.hpp as of commit 8cafafa (round 4: entries chain | lcp << 16)
#   hc_new.so      the library of the working tree (entries carry the mismatch byte)
#   wave_old.so / wave_new.so (tools/r05/wave_ab.sh): the working tree's library / the same with tools/ab/wave_decoder_wider_bursts.patch applied
#   hc_new2.so (profiles/r05/hc_head_entry_register_window_ab.txt): with tools/ab/hc_head_entry_register_window.patch applied
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc -DLZ4HIP_TUNING_BUILD -DLZ4HIP_DEC4_DUAL_STORE=0 lz4net_amd/csrc/lz4hip_api.hip -o build_variants/tune_base.so
/opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc -DLZ4HIP_TUNING_BUILD -DLZ4HIP_DEC4_DUAL_STORE=2 lz4net_amd/csrc/lz4hip_api.hip -o build_variants/tune_dual2.so
/opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc lz4net_amd/csrc/lz4hip_api.hip -o build_variants/hc_new.so
T=$(mktemp -d); mkdir -p $T/lz4net_amd $T/tools; cp -r lz4net_amd/csrc $T/lz4net_amd/; cp -r include $T/; cp -r tools/ab $T/tools/
git show 8cafafa:lz4net_amd/csrc/lz4hip_hc_lcp.hpp > $T/lz4net_amd/csrc/lz4hip_hc_lcp.hpp
/opt/rocm/bin/hipcc $F -I$T/lz4net_amd/csrc $T/lz4net_amd/csrc/lz4hip_api.hip -o build_variants/hc_old.so
rm -rf $T
cp build_variants/hc_new.so build_variants/wave_old.so
# (a clean working tree is assumed: each patch is applied, built and reverted in place)
for v in wave_new:wave_decoder_wider_bursts hc_new2:hc_head_entry_register_window; do
  git apply tools/ab/${v#*:}.patch && /opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc lz4net_amd/csrc/lz4hip_api.hip -o build_variants/${v%%:*}.so; git apply -R tools/ab/${v#*:}.patch
done
ls -la build_variants
