#!/bin/bash
# The prebuilt library variants tools/r05/call1.sh / call2.sh / call5.sh copy over lz4net_amd/liblz4hip.so (run HERE, in the build container, before
# the gpurun call: hipcc cross-compiles; build_variants/ is git-ignored but travels to the GPU box).
#   tune_base.so   tuning build (every lane-decoder configuration), ring rows wrapped (round 4's appends)
#   tune_dual2.so  tuning build, ring rows stored twice (ds_write2st64_b32), the round-5 default
#   hc_old.so      the library with lz4hip_hc_lcp.hpp as of commit 8cafafa (round 4: entries chain | lcp << 16)
#   hc_new.so      the library of the working tree (entries carry the mismatch byte)
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
ls -la build_variants
