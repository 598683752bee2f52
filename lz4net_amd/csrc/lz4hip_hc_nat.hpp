// lz4hip_hc_nat.hpp -- batched LZ4HC for blocks <= 64 KiB, one LANE per block like lz4hip_hc_conv.hpp and bit-exact to the
// same reference functions (LZ4_compressHCCtx and its match finder, original/lz4hc.c:330-755), WITHOUT the insert loop.
//
// What the counters said about lz4hip_hc_conv.hpp (profiles/r03/pmc_traffic.json): 25.9 MB of line traffic per 64 KiB block,
// a third of it LZ4HC_Insert (lz4hc.c:358-373) -- 65 k read-modify-writes of a bucket head per block, each one a line in and a
// line out because with a quarter of a million lanes in flight nothing survives in a cache.  But what the insert loop computes
// does not depend on the parse: every position of a block is inserted exactly once, in position order, before any search looks
// past it, so at the time position p is inserted its bucket's head is simply "the latest earlier position with the same hash"
// (position 0 for an empty bucket, lz4hc.c:332), and
//      chain[p] = p - (latest q < p with hash(q) == hash(p), else 0)            (the NATURAL chain)
// can be computed for all positions of all blocks up front, by a kernel that keeps the heads in LDS (hc_nat_chain_kernel: one
// wavefront per block, 64 consecutive positions per step, coalesced).  The same table also answers the one question the parse
// asks the heads -- the first candidate of a search at ip is ip - chain[ip], read BEFORE ip is inserted -- so the lane kernel
// needs no heads at all, no zero-fill, no input window, no chain stores.
//
// The repeat optimisation (lz4hc.c:437-455) is the one place where the reference's tables depend on the parse: after a search
// at ip that found its bucket's head within 4 bytes (delta = ip - head) matching `repl` bytes, the positions R = [ip, end),
// end = ip + repl - 3, get chain = delta instead of being inserted one by one, and only the last `delta` of them are written
// to the heads.  Two facts make the natural chain exact nevertheless:
//  (1) R's chain entries: the lane kernel writes chain[q] = delta for q in R like the reference does (kHsRepl).  Positions
//      outside every R keep their natural entry, and that IS what the reference's insert computes for them, by (2).
//  (2) the heads never differ from "latest earlier position with the same hash" when anybody reads them.  A position q in
//      S = [ip, end - delta) is never written to the heads, but its 4-byte word equals the word at q + delta (both lie inside
//      the matched region: q + 3 <= end + 2 = ip + repl - 1), and q + delta is again in R; so some q + k*delta in the last
//      period [end - delta, end), which IS written to the heads, has the same hash and is later than q.  Every reader of a head
//      -- the insert of a position >= end, the search at a position >= end + 1 (the next search is at ip + ml - 2 with
//      ml >= repl, lz4hc.c:594) -- comes after `end`, so the latest same-hash position it sees is never one of S.
// Blocks > 64 KiB (32-bit heads, chain slots that wrap) stay with lz4hip_hc_conv.hpp.
#pragma once
#include "lz4hip_hc_conv.hpp"

namespace lz4hip {

constexpr size_t kHcNatChainBytes = 65536 * sizeof(uint16_t);   // per block
constexpr int kHcNatLdsBytes = 32768 * sizeof(uint16_t);        // heads of the chain builder: exactly 64 KiB (two workgroups per CU)
constexpr int kHcNatAhead = 8;                                  // steps whose input words are in flight

// The natural chain of blocks [first, first + gridDim.x): chains + k * kHcNatChainBytes is the table of block first + k.
// One workgroup of FOUR wavefronts per block, heads in LDS (64 KiB: two blocks = eight wavefronts per CU).  A step is 64
// consecutive positions; wavefront w takes the steps s = w, w + 4, ...  What a step costs is finding the lanes that share a
// bucket (sixteen ballots, ~120 of its ~140 vector instructions) and that needs no heads, so the wavefronts do it side by side;
// only the short head phases -- read the bucket's old head, write the new one -- take turns under LDS-only barriers, one
// per step (one wavefront per block was bound by its own instruction issue on one SIMD of four: 152 ms per 2^18 blocks; two
// wavefronts: 132 ms).
constexpr int kHcNatChainWaves = 4;
constexpr int kHcNatChainThreads = 64 * kHcNatChainWaves;
template <class EntryT>          // uint16_t: the chain alone; uint32_t: room for lz4hip_hc_lcp.hpp's length byte (written as 0 here)
__global__ void __launch_bounds__(kHcNatChainThreads) hc_nat_chain_kernel(Batch b, long long first, uint8_t* chains)
{
    LZ4HIP_DYN_LDS(lds);
    uint16_t* const head = (uint16_t*)lds;
    const int lane = wv::lane();
    const int w = wv::wave_in_block();
    const int64_t blk = (int64_t)first + blockIdx.x;
    const int n = wv::uniform(batch_src_len(b, blk));
    if (n > 65536) return;                                           // (the lane kernel reports it)
    const uint8_t* const in = batch_src(b, blk);
    EntryT* const chain = (EntryT*)(chains + (size_t)blockIdx.x * 65536 * sizeof(EntryT));
    for (int i = (int)threadIdx.x * 16; i < kHcNatLdsBytes; i += kHcNatChainThreads * 16) wv::store16(lds + i, 0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) chain[0] = 0xFFFF;                         // never inserted: DELTANEXT(base) keeps its initial value (lz4hc.c:333)
    const int last = n - 4;                                          // last position that has a 4-byte word
    const int steps = last >= 1 ? (last + 63) / 64 : 0;              // step s: positions 1 + 64 s + lane
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    // The input words of a wavefront's next kHcNatAhead steps are in flight while the current ones are processed (addresses
    // clamped to the block, no branches: the compiler can count them instead of waiting for all of them).
    constexpr int U = kHcNatAhead;
    auto word_of = [&](int s) -> uint32_t { const int p = 1 + s * 64 + lane; return load_u32(in + (p <= last ? p : (last >= 0 ? last : 0))); };
    constexpr int W = kHcNatChainWaves;
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = n >= 4 ? word_of(w + W * u) : 0u;
    for (int j0 = 0; W * j0 < steps; j0 += U) {                      // (round j: steps W j .. W j + W - 1; every wavefront runs every round)
#pragma unroll
        for (int u = 0; u < U; u++) nxt[u] = word_of(w + W * (j0 + U + u));
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (W * (j0 + u) >= steps) break;                        // (uniform over the workgroup)
            const int s = W * (j0 + u) + w;
            const int base = 1 + s * 64, p = base + lane;
            const bool active = p <= last;
            // ---- the lanes of my step that share my bucket (no heads needed) ----
            const uint32_t h = active ? hash15(cur[u]) : 0x8000u + (uint32_t)lane;   // (lanes past the end: a bucket of their own, never written)
            uint64_t diff = 0;                                       // lanes whose hash differs from mine in some bit
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t mine = (uint32_t)((int32_t)(h << (31 - k)) >> 31);   // bit k of my hash: 0 or ~0 (v_bfe_i32)
                diff |= wv::ballot(mine != 0u) ^ (((uint64_t)mine << 32) | mine);
            }
            const uint64_t same = ~diff;
            const uint64_t below = same & lanes_below;
            const bool is_last = ((same >> lane) >> 1) == 0;
            // ---- the head phases, in step order: each position chains to the nearest lower lane of its bucket, the lowest to
            //      the head read before the step; the highest becomes the head ----
            int old = 0;
#pragma unroll
            for (int turn = 0; turn < W; turn++) {
                wv::lds_barrier();                                   // the previous step's head phase is complete
                if (w == turn) { old = (int)head[h & 0x7FFFu]; asm volatile("" ::: "memory"); if (active & is_last) head[h] = (uint16_t)p; }
            }
            const int prev = below ? base + (63 - __builtin_clzll(below)) : old;
            if (active) chain[p] = (EntryT)(p - prev);
        }
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
}

// (the lane kernel that worked on these chains without shared lengths -- the generation lz4hip_hc_lcp.hpp replaced -- lives in
//  tools/ab/lz4hip_hc_nat_lane.hpp: A/B runs and emulator tests only)

}  // namespace lz4hip
