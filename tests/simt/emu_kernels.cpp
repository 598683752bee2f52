// emu_kernels.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the real kernel sources (lz4net_amd/csrc/*.hpp) against the SIMT emulator and exposes
// them through a C ABI for tests/test_simt_emulation.py.  Built with g++, never shipped.
#include "simt_wave.hpp"

static unsigned long long g_iterations = 0;   // loop iterations of the lane decoders (all wavefronts), counted by lane 0
#define LZ4HIP_ITERATION_HOOK(lane) do { if ((lane) == 0) g_iterations++; } while (0)
static unsigned long long g_stat[32];         // lane-iterations per state of the fourth-generation lane decoder (tools/emu_decoder_stats.py)
#define LZ4HIP_STAT(slot, cond) do { if (cond) g_stat[slot]++; } while (0)
#define LZ4HIP_STAT_ADD(slot, n) do { g_stat[slot] += (unsigned long long)(n); } while (0)

#include "lz4hip_common.hpp"
#include "lz4hip_decode.hpp"
#include "lz4hip_decode_lane.hpp"
#include "lz4hip_decode_lane3.hpp"
#include "lz4hip_decode_lane4.hpp"
#include "lz4hip_encode.hpp"
#include "lz4hip_encode_lane.hpp"
#include "lz4hip_synth.hpp"
#ifdef LZ4HIP_HAVE_HC
#include "lz4hip_hc.hpp"
#include "lz4hip_hc_lane.hpp"
#include "lz4hip_hc_conv.hpp"
#include "lz4hip_hc_nat.hpp"
#include "lz4hip_hc_nat_lane.hpp"
#include "lz4hip_hc_lcp.hpp"
#endif

using namespace lz4hip;

static Batch make_batch(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                        int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n)
{
    Batch b;
    memset(&b, 0, sizeof b);
    b.src = src; b.src_stride = src_stride; b.src_len = src_len;
    b.dst = dst; b.dst_stride = dst_stride; b.dst_cap = dst_cap;
    b.result = result; b.n_blocks = n;
    return b;
}

extern "C" {

void emu_decode(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int waves_per_group, int filter)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    const unsigned wpg = (unsigned)waves_per_group;
    dim3 grid((unsigned)((n + wpg - 1) / wpg)), block(64 * wpg);
    const size_t lds = (size_t)wpg * kWaveLdsBytes;
    if (known) simt::launch(grid, block, lds, [=] { decode_kernel<true>(b, filter); });
    else       simt::launch(grid, block, lds, [=] { decode_kernel<false>(b, filter); });
}

void emu_decode_lane(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                     int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int filter, int ring, int stage)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define EMU_LANE(R, SB)                                                                                                        \
    do {                                                                                                                       \
        if (known) simt::launch(grid, block, lane_decode_lds_bytes(R, SB), [=] { decode_lane_kernel<true, R, SB>(b, filter); });  \
        else       simt::launch(grid, block, lane_decode_lds_bytes(R, SB), [=] { decode_lane_kernel<false, R, SB>(b, filter); }); \
    } while (0)
    if (ring == 128 && stage == 64) EMU_LANE(128, 64);
    else if (ring == 128) EMU_LANE(128, 128);
    else if (stage == 64) EMU_LANE(256, 64);
    else EMU_LANE(256, 128);
#undef EMU_LANE
}

// third-generation lane decoder (lz4hip_decode_lane3.hpp); `ring` = bytes of output ring per lane
void emu_decode_lane3(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                      int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int filter, int ring, int stage)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define EMU_LANE3(R, SB)                                                                                                        \
    do {                                                                                                                        \
        if (known) simt::launch(grid, block, lane3_lds_bytes(R, SB), [=] { decode_lane3_kernel<true, R, SB>(b, filter); });        \
        else       simt::launch(grid, block, lane3_lds_bytes(R, SB), [=] { decode_lane3_kernel<false, R, SB>(b, filter); });       \
    } while (0)
    if (ring == 128) EMU_LANE3(128, 64);
    else if (ring == 176) EMU_LANE3(176, 64);
    else if (ring == 240) EMU_LANE3(240, 64);
    else if (ring == 256 && stage == 128) EMU_LANE3(256, 128);
    else if (ring == 256) EMU_LANE3(256, 64);
    else simt::die("emu_decode_lane3: ring size not instantiated", ring, stage);
#undef EMU_LANE3
}

// fourth-generation lane decoder (lz4hip_decode_lane4.hpp); cfg = ring bytes + 1000 x variant (bit 0: 128-byte flush units, bit 1: 32-byte pieces)
void emu_decode_lane4(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                      int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int filter, int cfg)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define EMU_LANE4(CFG)                                                                                                          \
    case CFG: {                                                                                                                 \
        constexpr int R = (CFG) % 1000, P = ((CFG) / 1000 & 2) ? 32 : 64, FU = ((CFG) / 1000 & 1) ? 128 : 64, FS = ((CFG) / 1000 & 4) ? 1 : 2, FE = ((CFG) / 1000 & 8) ? 2 : 1, IE = ((CFG) / 1000 & 16) ? 2 : 1, POL = ((CFG) / 1000 & 32) ? 16 : 0;  \
        if (known) simt::launch(grid, block, lane4_lds_bytes(R), [=] { decode_lane4_kernel<true, R, P, FU, FS, FE, IE, POL>(b, filter); });     \
        else       simt::launch(grid, block, lane4_lds_bytes(R), [=] { decode_lane4_kernel<false, R, P, FU, FS, FE, IE, POL>(b, filter); });    \
    } break
    switch (cfg) {
    EMU_LANE4(128); EMU_LANE4(2128); EMU_LANE4(192); EMU_LANE4(1192); EMU_LANE4(3192); EMU_LANE4(7192); EMU_LANE4(1256); EMU_LANE4(2240); EMU_LANE4(5256); EMU_LANE4(11192); EMU_LANE4(15192); EMU_LANE4(27192); EMU_LANE4(25192); EMU_LANE4(59192); EMU_LANE4(35192); EMU_LANE4(34128);
    default: simt::die("emu_decode_lane4: configuration not instantiated", cfg, 0);
    }
#undef EMU_LANE4
}

// the default configuration in workgroups of FOUR wavefronts (decode_lane4_wg4_kernel: the four rings interleaved across 256 lanes); wrapped != 0: POL bit 5
void emu_decode_lane4_wg4(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                          int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int filter, int wrapped)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const size_t lds = 4u * lane4_lds_bytes(192);
    if (wrapped) {
        if (known) simt::launch(grid, block, lds, [=] { decode_lane4_wg4_kernel<true, 192, 32, 128, 2, 2, 2, 16 | 32>(b, filter); });
        else       simt::launch(grid, block, lds, [=] { decode_lane4_wg4_kernel<false, 192, 32, 128, 2, 2, 2, 16 | 32>(b, filter); });
    } else {
        if (known) simt::launch(grid, block, lds, [=] { decode_lane4_wg4_kernel<true, 192, 32, 128, 2, 2, 2, 16>(b, filter); });
        else       simt::launch(grid, block, lds, [=] { decode_lane4_wg4_kernel<false, 192, 32, 128, 2, 2, 2, 16>(b, filter); });
    }
}

// the persistent variant of the default configuration: `groups` wavefronts whose lanes pull blocks from a counter
void emu_decode_lane4_persistent(int known, const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                                 int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int filter, int groups)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    static unsigned long long counter;
    counter = 0;
    unsigned long long* c = &counter;
    dim3 grid((unsigned)groups), block(64);
    if (known) simt::launch(grid, block, lane4_lds_bytes(192), [=] { decode_lane4_persistent_kernel<true, 192, 32, 128, 2, 2, 2, 16>(b, filter, c); });
    else       simt::launch(grid, block, lane4_lds_bytes(192), [=] { decode_lane4_persistent_kernel<false, 192, 32, 128, 2, 2, 2, 16>(b, filter, c); });
}

void emu_encode_fast(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                     int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    simt::launch(dim3((unsigned)n), dim3(64), kFastTableBytes, [=] { encode_fast_kernel(b, 0); });
}

// launch_encode's form for batches of more than nine blocks per CU: five blocks per workgroup, each wavefront with its own 16 KiB of the allocation
void emu_encode_fast_wg5(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                         int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    simt::launch(dim3((unsigned)((n + 4) / 5)), dim3(64 * 5), 5 * kFastTableBytes, [=] { encode_fast_kernel<2, 5>(b, 0); });
}

#ifdef LZ4HIP_HAVE_HC
void emu_encode_hc(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                   int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups, int heads32)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    // chain slabs poisoned with a pattern that would derail any walk reading an unwritten slot
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)groups * kHcGlobalBytesPerGroup, 0x5A);
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    uint8_t* chains = ws.data() + 256;
    const int lds = heads32 ? kHcLdsHeads32 : kHcLdsHeads16;
    simt::launch(dim3((unsigned)groups), dim3(64), (size_t)lds, [=] { encode_hc_kernel(b, counter, chains, lds); });
}
#endif

void emu_encode_fast_lane(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                          int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    // the slab as the library builds it: equally sized chunks, a wavefront's tables never split (here: two wavefronts per chunk, chunks out of order)
    const unsigned tables_per_chunk = 128;
    const int n_chunks = (groups * 64 + (int)tables_per_chunk - 1) / (int)tables_per_chunk;
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)n_chunks * tables_per_chunk * kLaneTableBytes, 0x5A);     // poisoned: the kernel must zero its tables
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    static std::vector<uint8_t*> chunks;
    chunks.clear();
    for (int c = 0; c < n_chunks; c++) chunks.push_back(ws.data() + 256 + (size_t)(n_chunks - 1 - c) * tables_per_chunk * kLaneTableBytes);
    uint8_t* const* cp = chunks.data();
    simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_fast_lane_kernel(b, counter, cp, tables_per_chunk, 0); });
}

// The two launches of a large batch (launch_encode, 'a'): one wavefront per block with the hand-over rule, then one lane
// per block over the blocks handed over.  `deferred` receives 1 for every block the first launch handed over.
void emu_encode_fast_two_launches(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                                  int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int32_t* deferred)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    simt::launch(dim3((unsigned)n), dim3(64), kFastTableBytes, [=] { encode_fast_kernel(b, kEncodeMayDefer); });
    for (int64_t i = 0; i < n; i++) deferred[i] = result[i] == kDeferredResult;
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)64 * kLaneTableBytes, 0x5A);
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    static uint8_t* one_chunk[1];
    one_chunk[0] = ws.data() + 256;
    uint8_t* const* cp = one_chunk;
    simt::launch(dim3(1), dim3(64), 0, [=] { encode_fast_lane_kernel(b, counter, cp, 64u, 1); });
}

#ifdef LZ4HIP_HAVE_HC
void emu_encode_hc_lane(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                        int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups, int heads32)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    const size_t slab = heads32 ? kHcLaneSlab32 : kHcLaneSlab16;
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)groups * 64 * slab, 0x5A);      // poisoned
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    uint8_t* slabs = ws.data() + 256;
    simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_hc_lane_kernel(b, counter, slabs, (unsigned long long)slab); });
}
#endif

#ifdef LZ4HIP_HAVE_HC
// Slab reuse of the LZ4HC lane kernel, deterministically: lane L of the single wavefront encodes blocks L, L + 64, L + 128, ...
// IN THAT ORDER on its own slab (the product kernel hands blocks out through an atomic counter, so which lane gets which
// block is not reproducible; the per-block function and the slab are the product's).
static void hc_lane_static_kernel(Batch b, uint8_t* slabs, unsigned long long slab_bytes)
{
    uint8_t* slab = slabs + (size_t)threadIdx.x * (size_t)slab_bytes;
    for (int64_t blk = threadIdx.x; blk < b.n_blocks; blk += 64) {
        const int n = batch_src_len(b, blk), cap = batch_dst_cap(b, blk);
        int r;
        if (n <= 65536) r = lane_encode_hc_block<uint16_t>(batch_src(b, blk), n, batch_dst(b, blk), cap, slab);
        else            r = lane_encode_hc_block<uint32_t>(batch_src(b, blk), n, batch_dst(b, blk), cap, slab);
        b.result[blk] = r;
    }
}
void emu_encode_hc_lane_static(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                               int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    static std::vector<uint8_t> ws;
    ws.assign((size_t)64 * kHcLaneSlab32, 0x5A);                    // poisoned once; afterwards whatever the previous block left
    uint8_t* slabs = ws.data();
    simt::launch(dim3(1), dim3(64), 0, [=] { hc_lane_static_kernel(b, slabs, (unsigned long long)kHcLaneSlab32); });
}
#endif

#ifdef LZ4HIP_HAVE_HC
// convergent lane-per-block LZ4HC (lz4hip_hc_conv.hpp): same launch shape as emu_encode_hc_lane
void emu_encode_hc_conv(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                        int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups, int heads32)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    const size_t slab = heads32 ? kHcLaneSlab32 : kHcLaneSlab16;
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)groups * 64 * slab, 0x5A);      // poisoned
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    uint8_t* slabs = ws.data() + 256;
    if (heads32) simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_hc_conv_kernel<uint32_t>(b, counter, slabs, (unsigned long long)slab); });
    else         simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_hc_conv_kernel<uint16_t>(b, counter, slabs, (unsigned long long)slab); });
}
#endif

#ifdef LZ4HIP_HAVE_HC
// LZ4HC without the insert loop (lz4hip_hc_nat.hpp): the chain builder over every block, then the lane kernel
void emu_encode_hc_nat(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                       int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)n * kHcNatChainBytes, 0x5A);      // poisoned
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    uint8_t* chains = ws.data() + 256;
    simt::launch(dim3((unsigned)n), dim3(kHcNatChainThreads), kHcNatLdsBytes, [=] { hc_nat_chain_kernel<uint16_t>(b, 0, chains); });
    simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_hc_nat_kernel(b, 0, (long long)n, counter, chains); });
}
#endif

#ifdef LZ4HIP_HAVE_HC
// LZ4HC over chains that carry shared lengths (lz4hip_hc_lcp.hpp): chain builder, length fill, lane kernel
void emu_encode_hc_lcp(const uint8_t* src, int64_t src_stride, const int32_t* src_len, uint8_t* dst,
                       int64_t dst_stride, const int32_t* dst_cap, int32_t* result, int64_t n, int groups)
{
    Batch b = make_batch(src, src_stride, src_len, dst, dst_stride, dst_cap, result, n);
    static std::vector<uint8_t> ws;
    ws.assign(256 + (size_t)n * kHcLcpTableBytes, 0x5A);      // poisoned
    memset(ws.data(), 0, 256);
    unsigned long long* counter = (unsigned long long*)ws.data();
    uint8_t* tables = ws.data() + 256;
    simt::launch(dim3((unsigned)n), dim3(kHcNatChainThreads), kHcNatLdsBytes, [=] { hc_nat_chain_kernel<uint32_t>(b, 0, tables); });
    simt::launch(dim3((unsigned)n), dim3(kHcLcpFillThreads), kHcLcpFillLdsBytes, [=] { hc_lcp_fill_kernel(b, 0, tables); });
    simt::launch(dim3((unsigned)groups), dim3(64), 0, [=] { encode_hc_lcp_kernel(b, 0, (long long)n, counter, tables, kHcLcpCtrlEvery, kHcLcpCtrlLanes); });
}
#endif

void emu_synth(int dist, uint64_t seed, uint64_t first_block, uint64_t block_step, int64_t n, uint8_t* out, int64_t stride, int len)
{
    SynthArgs a = { out, stride, n, seed, first_block, block_step, len, dist };
    unsigned grid = (dist <= 1) ? 3u : (unsigned)((n + 63) / 64);
    simt::launch(dim3(grid), dim3(64), 0, [=] { synth_kernel(a); });
}

void emu_checksum(const uint8_t* data, int64_t stride, const int32_t* len, uint64_t* sums, int64_t n)
{
    ChecksumArgs a = { data, nullptr, stride, len, 0, sums, n };
    simt::launch(dim3((unsigned)((n + 3) / 4)), dim3(256), 0, [=] { checksum_kernel(a); });
}

unsigned long long emu_compare(const uint8_t* x, int64_t xs, const uint8_t* y, int64_t ys, const int32_t* len, int64_t n)
{
    unsigned long long bad = 0;
    CompareArgs c = { x, xs, y, ys, len, 0, n, &bad };
    simt::launch(dim3((unsigned)((n + 3) / 4)), dim3(256), 0, [=] { compare_kernel(c); });
    return bad;
}

unsigned long long emu_steps() { return simt::rt().steps; }
void emu_stats(unsigned long long* out, int reset) { for (int i = 0; i < 32; i++) { out[i] = g_stat[i]; if (reset) g_stat[i] = 0; } }
unsigned long long emu_iterations(int reset) { const unsigned long long v = g_iterations; if (reset) g_iterations = 0; return v; }
}
