#!/bin/bash
# Prebuilt library variants for same-box A/B runs (run HERE, in the build container, before the gpurun call; build_variants/ is git-ignored but
# travels to the GPU box; the A/B scripts copy a variant over lz4net_amd/liblz4hip.so and run with LZ4HIP_KEEP_LIBRARY=1 so that build.py keeps it):
#   wave_dec_old.so  the working tree with lz4hip_decode.hpp as of commit 84019f8 (bursts: <= 6 literals, no length byte)
#   wave_dec_new.so  the working tree (bursts: <= 14 literals, one match-length byte)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc lz4net_amd/csrc/lz4hip_api.hip -o build_variants/wave_dec_new.so
T=$(mktemp -d); mkdir -p $T/lz4net_amd $T/tools; cp -r lz4net_amd/csrc $T/lz4net_amd/; cp -r include $T/; cp -r tools/ab $T/tools/
git show 84019f8:lz4net_amd/csrc/lz4hip_decode.hpp > $T/lz4net_amd/csrc/lz4hip_decode.hpp
/opt/rocm/bin/hipcc $F -I$T/lz4net_amd/csrc $T/lz4net_amd/csrc/lz4hip_api.hip -o build_variants/wave_dec_old.so
rm -rf $T
ls -la build_variants
#   enc_wave_a.so    the working tree with lz4hip_encode.hpp as of commit 84019f8 (second version of the 64k encoder, no prefetches)
#   enc_wave_b.so    the working tree (the next search's first-step words and the next sequence's literals requested ahead)
/opt/rocm/bin/hipcc $F -Ilz4net_amd/csrc lz4net_amd/csrc/lz4hip_api.hip -o build_variants/enc_wave_b.so
T=$(mktemp -d); mkdir -p $T/lz4net_amd $T/tools; cp -r lz4net_amd/csrc $T/lz4net_amd/; cp -r include $T/; cp -r tools/ab $T/tools/
git show 84019f8:lz4net_amd/csrc/lz4hip_encode.hpp > $T/lz4net_amd/csrc/lz4hip_encode.hpp
/opt/rocm/bin/hipcc $F -I$T/lz4net_amd/csrc $T/lz4net_amd/csrc/lz4hip_api.hip -o build_variants/enc_wave_a.so
rm -rf $T
ls -la build_variants
