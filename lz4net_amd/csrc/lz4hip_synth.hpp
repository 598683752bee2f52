// lz4hip_synth.hpp -- device-side synthetic block generators, block checksums and buffer compare.
//
// BASELINE configs 2/3/5 are 2^20..2^23 blocks of 64 KiB (64..512 GiB): the batch is generated,
// encoded, decoded and verified entirely in HBM.  The generators are bit-identical to the CPU
// twins in oracle/synth.c (tests/test_synth_parity.py), so any block of a full-size batch can be
// regenerated on the host and pushed through the oracle for a spot check.
//
//   0 zeros | 1 incompressible (counter-based splitmix64) | 2 the reference's own fuzzer generator
//   (original/fuzzer.c:81-85,149-168) seeded per block | 3 record-like (long matches)
//
// One LANE per block for the serially-dependent generators (2, 3); bytes are packed into 16-byte
// stores.  This is set-up work, not part of any timed region.
#pragma once
#include "lz4hip_common.hpp"

namespace lz4hip {

LZ4HIP_DEVICE uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
LZ4HIP_DEVICE uint64_t block_key(uint64_t seed, uint64_t block) { return mix64(mix64(seed) + block); }
LZ4HIP_DEVICE uint32_t lcg(uint32_t& s) { s = s * 2654435761u + 2246822519u; return s; }   // FUZ_rand

struct SynthArgs {
    uint8_t* out;
    int64_t stride;
    int64_t n_blocks;
    uint64_t seed;
    uint64_t first_block;
    uint64_t block_step;        // block i of the batch is synthetic block first_block + i * block_step
    int32_t len;
    int32_t dist;
};

// 16-byte staging so that one lane emits aligned-ish wide stores instead of 64 Ki byte stores.
struct BytePacker {
    uint8_t* out; int pos; Vec16 acc;
    LZ4HIP_DEVICE void init(uint8_t* o) { out = o; pos = 0; acc = Vec16{ { 0, 0, 0, 0 } }; }
    LZ4HIP_DEVICE void put(uint8_t b)
    {
        const int k = pos & 15;
        acc.w[k >> 2] |= (uint32_t)b << ((k & 3) * 8);
        pos++;
        if ((pos & 15) == 0) { store_v16(out + pos - 16, acc); acc = Vec16{ { 0, 0, 0, 0 } }; }
    }
    LZ4HIP_DEVICE void flush()
    {
        const int k = pos & 15;
        for (int i = 0; i < k; i++) out[pos - k + i] = (uint8_t)(acc.w[i >> 2] >> ((i & 3) * 8));
    }
};

__global__ void __launch_bounds__(64) synth_kernel(SynthArgs a)
{
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.dist == 0 || a.dist == 1) {
        // fully parallel: every thread writes 8-byte words, grid-strided over the whole batch
        const int64_t words_per_block = (a.len + 7) / 8;
        const int64_t total = words_per_block * a.n_blocks;
        for (int64_t w = blk; w < total; w += (int64_t)gridDim.x * blockDim.x) {
            const int64_t bi = w / words_per_block, wi = w % words_per_block;
            uint64_t v = 0;
            if (a.dist == 1) v = mix64(block_key(a.seed, a.first_block + (uint64_t)bi * a.block_step) + (uint64_t)wi * 0xD1342543DE82EF95ull);
            uint8_t* p = a.out + bi * a.stride + wi * 8;
            const int room = a.len - (int)(wi * 8);
            if (room >= 8) __builtin_memcpy(p, &v, 8);
            else for (int k = 0; k < room; k++) p[k] = (uint8_t)(v >> (8 * k));
        }
        return;
    }
    if (blk >= a.n_blocks) return;
    uint8_t* out = a.out + blk * a.stride;
    const uint64_t key = block_key(a.seed, a.first_block + (uint64_t)blk * a.block_step);
    if (a.dist == 2) {
        // original/fuzzer.c:149-168
        BytePacker pk; pk.init(out);
        uint32_t s = (uint32_t)key, seeds[4], cur = 3266489917u;
        lcg(s);
        for (int j = 0; j < 4; j++) { seeds[j] = lcg(s) << 8; seeds[j] ^= (lcg(s) >> 8) & 65535u; }
        for (int j = 0; j < a.len; j++) {
            const int32_t k = (int32_t)lcg(s);
            if (j == 0 || ((k >> 10) % 10) == 0) {
                const uint32_t q = (lcg(s) >> 16) & 3u;
                cur = q == 0 ? seeds[0] : q == 1 ? seeds[1] : q == 2 ? seeds[2] : seeds[3];
            }
            if (((k >> 8) & 255) == 0) {
                const uint32_t q = (lcg(s) >> 16) & 3u;
                uint32_t v = lcg(s) << 8;
                v ^= (lcg(s) >> 8) & 65535u;
                if (q == 0) seeds[0] = v; else if (q == 1) seeds[1] = v; else if (q == 2) seeds[2] = v; else seeds[3] = v;
            }
            pk.put((uint8_t)(lcg(cur) >> 16));
        }
        pk.flush();
    } else {
        // record-like: oracle/synth.c fill_records
        uint32_t s = (uint32_t)(key >> 16) | 1u;
        uint32_t lit_left = 0, mat_left = 0, dist = 0;
        for (int pos = 0; pos < a.len; pos++) {
            uint8_t bt;
            if (lit_left == 0 && mat_left == 0) {
                const uint32_t r = lcg(s);
                lit_left = 4 + ((r >> 8) % 24);
                mat_left = pos > 0 ? 8 + ((r >> 16) % 88) : 0;
            }
            if (lit_left) {
                bt = (uint8_t)(0x20 + ((lcg(s) >> 16) & 63u));
                lit_left--;
                if (lit_left == 0 && mat_left) {
                    const uint32_t window = (uint32_t)(pos + 1) < 32768u ? (uint32_t)(pos + 1) : 32768u;
                    dist = 1 + ((lcg(s) >> 4) % window);
                }
            } else {
                bt = out[pos - (int)dist];
                mat_left--;
            }
            out[pos] = bt;
        }
    }
}

// Position-salted 64-bit checksum per block: sum over 8-byte little-endian words (tail zero padded)
// of mix64(word + index * K), plus mix64(len).  Order independent, so a wavefront sums it in
// parallel (device twin of lz4s_checksum in oracle/synth.c).  Used for "checksum of checksums"
// parity at full batch size, never in a timed region.
struct ChecksumArgs {
    const uint8_t* data; const int64_t* off; int64_t stride; const int32_t* len; int32_t len_all;
    uint64_t* sums; int64_t n_blocks;
};
__global__ void __launch_bounds__(256) checksum_kernel(ChecksumArgs a)
{
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv::wave_in_block();
    if (blk >= a.n_blocks) return;
    const uint8_t* p = a.data + (a.off ? a.off[blk] : blk * a.stride);
    const int n = wv::uniform(a.len ? a.len[blk] : a.len_all);
    const int lane = wv::lane();
    const int words = (n + 7) >> 3;
    uint64_t h = 0;
    for (int w = lane; w < words; w += 64) {
        uint64_t v = 0;
        const int room = n - w * 8;
        if (room >= 8) __builtin_memcpy(&v, p + (int64_t)w * 8, 8);
        else for (int k = 0; k < room; k++) v |= (uint64_t)p[(int64_t)w * 8 + k] << (8 * k);
        h += mix64(v + (uint64_t)w * 0xD1342543DE82EF95ull);
    }
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = wv::shuffle((uint32_t)h, lane ^ d), hi = wv::shuffle((uint32_t)(h >> 32), lane ^ d);
        h += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) a.sums[blk] = h + mix64((uint64_t)(uint32_t)n);
}

// Number of differing bytes between two strided batches (wave per block, 16 B per lane).
struct CompareArgs {
    const uint8_t* a; int64_t a_stride; const uint8_t* b; int64_t b_stride;
    const int32_t* len; int32_t len_all; int64_t n_blocks; unsigned long long* mismatches;
};
__global__ void __launch_bounds__(256) compare_kernel(CompareArgs c)
{
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv::wave_in_block();
    if (blk >= c.n_blocks) return;
    const uint8_t* pa = c.a + blk * c.a_stride;
    const uint8_t* pb = c.b + blk * c.b_stride;
    const int n = c.len ? c.len[blk] : c.len_all;
    const int lane = wv::lane();
    unsigned bad = 0;
    const int body = n & ~15;
    for (int k = lane * 16; k < body; k += 1024) {
        const Vec16 x = load_v16(pa + k), y = load_v16(pb + k);
        for (int i = 0; i < 4; i++) {
            uint32_t d = x.w[i] ^ y.w[i];
            bad += ((d & 0xFFu) != 0) + ((d & 0xFF00u) != 0) + ((d & 0xFF0000u) != 0) + ((d & 0xFF000000u) != 0);
        }
    }
    const int t = body + lane;
    if (t < n && pa[t] != pb[t]) bad++;
    if (bad) atomicAdd(c.mismatches, (unsigned long long)bad);
}

}  // namespace lz4hip
