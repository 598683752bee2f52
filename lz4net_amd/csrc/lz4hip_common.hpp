// lz4hip_common.hpp -- format constants and the batch descriptor shared by all lz4hip kernels.
//
// Constants follow the reference's block format and tuning values verbatim because they are baked
// into the compressed bytes (original/lz4.c:182-203,566-570; original/lz4hc.c:173-195;
// C# twins src/LZ4ps/LZ4Codec.cs:56-114).
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef LZ4HIP_WAVE_API
#error "include lz4hip_wave.hpp (or the test emulation of it) before any lz4hip kernel header"
#endif

namespace lz4hip {

constexpr int kMinMatch = 4;
constexpr int kLastLiterals = 5;        // last 5 bytes of a block are literals
constexpr int kMfLimit = 12;            // no match starts within the last 12 bytes
constexpr int kMinLength = 13;          // shorter inputs: one literal run
constexpr int kMaxDistance = 65535;
constexpr int k64kLimit = 65536 + 11;   // LZ4_64KLIMIT (original/lz4.c:566)
constexpr uint32_t kGolden = 2654435761u;
constexpr int kHcAttempts = 256;        // MAX_NB_ATTEMPTS (original/lz4hc.c:184)
constexpr int kHcOptimalMl = 18;        // OPTIMAL_ML (original/lz4hc.c:194)
constexpr int kFastTableBytes = 16384;  // u16[8192] (64k variant) or u32[4096] (generic variant)
// Fast encode of a large batch runs two launches: one wavefront per block first (lz4hip_encode.hpp), which hands a
// block whose sequences come thick and fast over to the lane-per-block launch by leaving this value in result[]
// (never a valid return value: sizes are >= 0, error codes > INT32_MIN).
constexpr int32_t kDeferredResult = INT32_MIN;
constexpr int kDeferCheckSequences = 16;   // every 16 sequences ...
constexpr int kDeferBytesPerSequence = 64;  // ... the block is handed over if they covered less than 16 x 64 input bytes

// One batch of independent blocks, device-resident.  Block i lives at base + (off ? off[i] : i*stride).
struct Batch {
    const uint8_t* src;
    const int64_t* src_off;     // optional explicit byte offsets (packed layouts); nullptr => i*src_stride
    int64_t src_stride;
    const int32_t* src_len;     // per-block input length in bytes
    uint8_t* dst;
    const int64_t* dst_off;
    int64_t dst_stride;
    const int32_t* dst_cap;     // per-block output capacity (encode / unknown-size decode) or exact size (known-size decode)
    int32_t dst_cap_all;        // used when dst_cap == nullptr
    int32_t src_len_all;        // used when src_len == nullptr
    int32_t* result;            // per-block return value, reference conventions (SURVEY.md 8b "Error conventions")
    int64_t n_blocks;
};

LZ4HIP_DEVICE const uint8_t* batch_src(const Batch& b, int64_t i) { return b.src + (b.src_off ? b.src_off[i] : i * b.src_stride); }
LZ4HIP_DEVICE uint8_t* batch_dst(const Batch& b, int64_t i) { return b.dst + (b.dst_off ? b.dst_off[i] : i * b.dst_stride); }
LZ4HIP_DEVICE int32_t batch_src_len(const Batch& b, int64_t i) { return b.src_len ? b.src_len[i] : b.src_len_all; }
LZ4HIP_DEVICE int32_t batch_dst_cap(const Batch& b, int64_t i) { return b.dst_cap ? b.dst_cap[i] : b.dst_cap_all; }

// Which blocks a decode launch handles.  A batch is decoded by two launches that partition it: blocks
// that are essentially long copies (compressed to < 1/8, or > 90 % literals) stream best with one
// wavefront per block, everything else (many short sequences) with one lane per block.
enum BlockFilter { kAllBlocks = 0, kStreamingBlocks = 1, kFineGrainedBlocks = 2 };
LZ4HIP_DEVICE bool block_selected(int filter, int src_len, int out_size)
{
    if (filter == kAllBlocks) return true;
    const bool streaming = (int64_t)src_len * 8 < out_size || (int64_t)src_len * 10 > (int64_t)out_size * 9;
    return streaming == (filter == kStreamingBlocks);
}

// Unaligned little-endian loads/stores (gfx950 runs with unaligned access enabled; hipcc lowers
// these to single global_load/store_dword[xN] instructions).
LZ4HIP_DEVICE uint32_t load_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
LZ4HIP_DEVICE uint64_t load_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
LZ4HIP_DEVICE void store_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
struct alignas(4) Vec16 { uint32_t w[4]; };
struct alignas(16) Aligned16 { uint32_t w[4]; };                 // 16-byte aligned LDS records
struct __attribute__((packed)) Packed16 { uint32_t w[4]; };      // 16 bytes at any address
LZ4HIP_DEVICE Vec16 load_v16(const uint8_t* p) { Vec16 v; wv::load16(p, v.w[0], v.w[1], v.w[2], v.w[3]); return v; }
LZ4HIP_DEVICE void store_v16(uint8_t* p, const Vec16& v) { wv::store16(p, v.w[0], v.w[1], v.w[2], v.w[3]); }

// Non-overlapping wave-cooperative copy of n bytes (literal runs: compressed stream <-> raw block).
// 16 bytes per lane per pass (1 KiB per wave-instruction), byte tail.
LZ4HIP_DEVICE void wave_copy(uint8_t* dst, const uint8_t* src, int n)
{
    const int lane = wv::lane();
    const int body = n & ~15;
    for (int k = lane * 16; k < body; k += 64 * 16) store_v16(dst + k, load_v16(src + k));
    const int t = body + lane;
    if (t < n) dst[t] = src[t];
}

// Wave-cooperative fill of n bytes with one value (runs of 255 in length encodings).
LZ4HIP_DEVICE void wave_fill(uint8_t* dst, uint8_t value, int n)
{
    for (int k = wv::lane(); k < n; k += 64) dst[k] = value;
}

}  // namespace lz4hip
