// lz4hip_encode_lane.hpp -- batched LZ4 fast block encoder for gfx950, one LANE per block
// (64 independent blocks per wavefront), bit-exact to the reference.
//
// Same functions as lz4hip_encode.hpp: LZ4_compress64kCtx (original/lz4.c:573-771 ==
// LZ4_compress64kCtx_64, src/LZ4pn/LZ4Codec.Unsafe64.Dirty.cs:303-528) below LZ4_64KLIMIT and
// LZ4_compressCtx (original/lz4.c:345-562) above it.
//
// Why a second mapping: the greedy parse is one chain of dependent hash-table and input reads per
// block (it has to be, to stay bit-exact), i.e. it is latency bound.  One wavefront per block with the
// table in LDS keeps at most 10 such chains in flight per CU; one lane per block keeps 64 per
// wavefront and >1000 per CU.  The price: the 16 KiB table of every resident lane lives in a global
// scratch slab (random 2-byte accesses), so this mapping is bound by scattered-access throughput
// instead of latency.  The grid is persistent (work handed out per lane by an atomic counter) so that
// the slab is sized by residency, not by the batch.
#pragma once
#include "lz4hip_common.hpp"

#include "lz4hip_encode.hpp"        // FastTable

namespace lz4hip {

constexpr int kLaneEncodeWavesPerCu = 16;     // twelve fresh processes (profiles/r04/encoder_reproducibility.txt): 16 -> 53.8 / 54.0 / 54.3 GB/s on D2, 24 -> 44.0 / 53.3 / 53.3
                                              // (round 3's 24 came from a same-process A/B: the rate is bimodal, 44-46 or 53-54, with where the slab lands)
constexpr int kLaneTableBytes = 32768;      // per lane: tagged u32[8192] (64k variant) or u32[4096] (generic variant)

// The hash table of one block as this mapping keeps it.  What the algorithm sees is exactly the reference's table
// (bucket -> last position inserted, empty bucket == position 0, lz4.c:583,642-651); the 64k variant additionally
// keeps, next to the position,
//   * 10 more bits of the inserted sequence's hash product: two sequences with different tags cannot be equal, so
//     most failing probes are decided without touching the input at the candidate position -- one far-away DRAM
//     sector less per probe, and distinct sectors are what bounds this kernel;
//   * a 6-bit epoch: an entry written for an earlier block of this lane is an empty bucket, so the table is zeroed
//     once per 63 blocks instead of once per block (32 KiB of stores per 64 KiB block otherwise).
template <bool GENERIC> struct LaneTable;
template <> struct LaneTable<true> {        // U32 HashTable[4096], positions as they are
    uint32_t* t; const uint8_t* in;
    LZ4HIP_DEVICE void clear(uint8_t* bytes, const uint8_t* input, int& epoch)
    {
        t = (uint32_t*)bytes; in = input;
        for (int k = 0; k < kFastTableBytes; k += 16) store_v16(bytes + k, Vec16{ { 0, 0, 0, 0 } });
        epoch = 63;                                                   // raw positions in the table: the 64k variant must zero it
    }
    LZ4HIP_DEVICE void put(uint32_t word, int pos) { t[FastTable<true>::hash(word)] = (uint32_t)pos; }
    // ref = table[h]; table[h] = pos; returns whether the 4 bytes at ref equal `word` (distance check included)
    LZ4HIP_DEVICE bool exchange(uint32_t word, int pos, int& ref)
    {
        const uint32_t h = FastTable<true>::hash(word);
        ref = (int)t[h];
        t[h] = (uint32_t)pos;
        if (ref < pos - kMaxDistance) return false;                   // lz4.c:427 / :538 (ref > ip - (MAX_DISTANCE + 1))
        return load_u32(in + ref) == word;
    }
};
template <> struct LaneTable<false> {       // U16 HashTable[8192] as (epoch:6 | tag:10 | position:16)
    uint32_t* t; const uint8_t* in; uint32_t word0, stamp;
    // `epoch` is the lane's counter across the blocks it encodes (1..63; 63 also means "contents unknown")
    LZ4HIP_DEVICE void clear(uint8_t* bytes, const uint8_t* input, int& epoch)
    {
        t = (uint32_t*)bytes; in = input; word0 = load_u32(input);     // an empty bucket is position 0 (never inserted)
        if (epoch >= 63) {
            for (int k = 0; k < kLaneTableBytes; k += 16) store_v16(bytes + k, Vec16{ { 0, 0, 0, 0 } });
            epoch = 0;
        }
        epoch++;
        stamp = (uint32_t)epoch << 26;
    }
    LZ4HIP_DEVICE void put(uint32_t word, int pos)
    {
        const uint32_t prod = word * kGolden;
        t[prod >> 19] = stamp | (((prod >> 4) & 0x3FFu) << 16) | (uint32_t)pos;
    }
    LZ4HIP_DEVICE bool exchange(uint32_t word, int pos, int& ref)
    {
        const uint32_t prod = word * kGolden, h = prod >> 19, tag = (prod >> 4) & 0x3FFu;
        const uint32_t e = t[h];
        t[h] = stamp | (tag << 16) | (uint32_t)pos;
        if ((e >> 26) != (stamp >> 26)) { ref = 0; return word0 == word; }   // empty bucket (or an older block's entry)
        ref = (int)(e & 0xFFFFu);
        if (((e >> 16) & 0x3FFu) != tag) return false;                // different sequences for certain
        return load_u32(in + ref) == word;
    }
};

// exact number of equal bytes in[a + i] == in[b + i] while a + i < limit (b < a)
LZ4HIP_DEVICE int lane_count_equal(const uint8_t* __restrict__ in, int a, int b, int limit)
{
    int n = 0;
    while (a + n + 16 <= limit) {                      // long runs (e.g. zero pages): 16 bytes per step
        const Vec16 x = load_v16(in + a + n), y = load_v16(in + b + n);
        const uint64_t d0 = (x.w[0] ^ y.w[0]) | ((uint64_t)(x.w[1] ^ y.w[1]) << 32);
        const uint64_t d1 = (x.w[2] ^ y.w[2]) | ((uint64_t)(x.w[3] ^ y.w[3]) << 32);
        if (d0) return n + (__builtin_ctzll(d0) >> 3);
        if (d1) return n + 8 + (__builtin_ctzll(d1) >> 3);
        n += 16;
    }
    while (a + n + 8 <= limit) {
        const uint64_t d = load_u64(in + a + n) ^ load_u64(in + b + n);
        if (d) return n + (__builtin_ctzll(d) >> 3);
        n += 8;
    }
    while (a + n < limit && in[a + n] == in[b + n]) n++;
    return n;
}

// The same count when the caller already holds the 16 bytes at in[a] (`x0`, valid if `x0_ok`: a + 16 <= limit): a
// search that measures many candidates against the same position loads its own side once.
LZ4HIP_DEVICE int lane_count_equal_from(const uint8_t* __restrict__ in, int a, int b, int limit, const Vec16& x0, bool x0_ok)
{
    if (!x0_ok) return lane_count_equal(in, a, b, limit);
    const Vec16 y = load_v16(in + b);
    const uint64_t d0 = (x0.w[0] ^ y.w[0]) | ((uint64_t)(x0.w[1] ^ y.w[1]) << 32);
    const uint64_t d1 = (x0.w[2] ^ y.w[2]) | ((uint64_t)(x0.w[3] ^ y.w[3]) << 32);
    if (d0) return __builtin_ctzll(d0) >> 3;
    if (d1) return 8 + (__builtin_ctzll(d1) >> 3);
    return 16 + lane_count_equal(in, a + 16, b + 16, limit);
}

// exact copy of n bytes (literal runs)
LZ4HIP_DEVICE void lane_copy(uint8_t* dst, const uint8_t* __restrict__ src, int n)
{
    int k = 0;
    for (; k + 16 <= n; k += 16) store_v16(dst + k, load_v16(src + k));
    if (k + 8 <= n) { store_u64(dst + k, load_u64(src + k)); k += 8; }
    for (; k < n; k++) dst[k] = src[k];
}

// length bytes after a saturated nibble: floor(rest/255) x 0xFF then rest % 255; returns bytes written
LZ4HIP_DEVICE int lane_put_length(uint8_t* out, int rest)
{
    int k = 0;
    for (; rest >= 255; rest -= 255) out[k++] = 255;
    out[k++] = (uint8_t)rest;
    return k;
}

template <bool GENERIC>
LZ4HIP_DEVICE int lane_encode_fast_block(const uint8_t* __restrict__ in, int n, uint8_t* out, int cap, uint8_t* table_bytes, int& epoch)
{
    LaneTable<GENERIC> table;
    const int mflimit = n - kMfLimit, matchlimit = n - kLastLiterals;
    int ip = 0, anchor = 0, op = 0;

    if (n >= kMinLength) {                                            // lz4.c:615
        table.clear(table_bytes, in, epoch);                          // fresh table (lz4.c:583); generic: HashTable[hash(0)] = 0 (lz4.c:403) == the fill
        ip = 1;                                                       // lz4.c:631
        // 8-byte register window over the input for the (mostly sequential) forward reads of the search loop
        uint64_t fw = load_u64(in + ip);                              // n >= 13: in-bounds
        int fw_pos = ip;
#define FWD_WORD(pos) ((uint32_t)(fw >> (8 * ((pos) - fw_pos))))
#define FWD_REFILL(pos)                                                                         \
        do { if ((pos) - fw_pos > 4 || (pos) < fw_pos) { fw_pos = (pos) + 8 <= n ? (pos) : n - 8; fw = load_u64(in + fw_pos); } } while (0)
        uint32_t fwd_word = FWD_WORD(ip);
        for (;;) {
            // ---- find a match: lz4.c:642-654 ----
            int attempts = 67, probe = ip, ref;
            uint32_t cur_word;
            bool out_of_input = false;
            for (;;) {
                cur_word = fwd_word;
                const int step = attempts++ >> 6;
                ip = probe;
                probe = ip + step;
                if (probe > mflimit) { out_of_input = true; break; }
                FWD_REFILL(probe);
                fwd_word = FWD_WORD(probe);
                if (table.exchange(cur_word, ip, ref)) break;       // lz4.c:649-654 (generic: distance test of :427 inside)
            }
            if (out_of_input) break;

            // ---- catch up: lz4.c:657 ----
            while (ip > anchor && ref > 0 && in[ip - 1] == in[ref - 1]) { ip--; ref--; }

            // ---- literals: lz4.c:660-691 ----
            int ll = ip - anchor;
            int token_at = op++;
            if (op + ll + (ll >> 8) > cap - 8) return 0;             // lz4.c:663
            if (ll >= 15 && op + (ll - 15) / 255 + 1 + ll > cap) return 0;   // (never write past cap; see lz4hip_encode.hpp)
            uint32_t token = ll >= 15 ? 0xF0u : (uint32_t)(ll << 4);
            // Short literal runs (the common case) are not copied now: token, literals and offset leave as ONE
            // 16-byte store once the match length -- the low nibble of the token -- is known.  The bytes of that
            // store beyond the offset are overwritten by whatever is emitted next.
            bool packed = ll <= 13 && token_at + 16 <= cap && anchor + 16 <= n;
            if (!packed) {
                if (ll >= 15) op += lane_put_length(out + op, ll - 15);
                lane_copy(out + op, in + anchor, ll);
            }
            op += ll;

            for (;;) {
                // ---- offset, match length: lz4.c:693-733 ----
                const uint32_t off = (uint32_t)(ip - ref) & 0xFFFFu;
                if (op + 2 > cap) return 0;
                if (!packed) { out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8); }
                op += 2;
                const int lit_from = anchor;
                ip += kMinMatch; ref += kMinMatch; anchor = ip;
                ip += lane_count_equal(in, ip, ref, matchlimit);
                const int extra = ip - anchor;
                if (op + (extra >> 8) > cap - 6) return 0;           // lz4.c:728
                if (extra >= 15 && op + (extra - 15) / 255 + 1 > cap) return 0;
                token |= extra >= 15 ? 15u : (uint32_t)extra;
                if (packed) {
                    uint64_t l0 = 0, l1 = 0;
                    if (ll > 0) { const Vec16 w = load_v16(in + lit_from); l0 = w.w[0] | ((uint64_t)w.w[1] << 32); l1 = w.w[2] | ((uint64_t)w.w[3] << 32); }
                    // keep ll literal bytes, shift them up by one byte, token below, offset above
                    if (ll < 8) { l0 &= (1ull << (8 * ll)) - 1ull; l1 = 0; }
                    else if (ll < 16) l1 &= (1ull << (8 * (ll - 8))) - 1ull;
                    uint64_t v0 = (l0 << 8) | token, v1 = (l1 << 8) | (l0 >> 56);
                    const int sh = 8 * (1 + ll);                     // 8 .. 112
                    if (sh < 64) { v0 |= (uint64_t)off << sh; v1 |= sh > 48 ? (uint64_t)off >> (64 - sh) : 0ull; }
                    else v1 |= (uint64_t)off << (sh - 64);
                    const Vec16 o = { { (uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32) } };
                    store_v16(out + token_at, o);
                } else {
                    out[token_at] = (uint8_t)token;
                }
                if (extra >= 15) op += lane_put_length(out + op, extra - 15);

                if (ip > mflimit) { anchor = ip; goto tail; }        // lz4.c:736
                // ---- re-seed the table and test the next position: lz4.c:739-751 ----
                fw_pos = ip - 2; fw = load_u64(in + fw_pos);           // ip <= mflimit: ip - 2 + 8 <= n
                table.put(FWD_WORD(ip - 2), ip - 2);
                cur_word = FWD_WORD(ip);
                if (!table.exchange(cur_word, ip, ref)) break;       // lz4.c:743-751 (generic: distance test of :538 inside)
                token_at = op++;                                      // zero-literal sequence (lz4.c:751)
                token = 0; ll = 0;
                packed = token_at + 16 <= cap;
            }
            anchor = ip++;                                            // lz4.c:754-755
            FWD_REFILL(ip);
            fwd_word = FWD_WORD(ip);
        }
#undef FWD_WORD
#undef FWD_REFILL
    }
tail:
    {   // ---- last literals: lz4.c:758-767 ----
        const int run = n - anchor;
        if (op + run + 1 + (run - 15 + 255) / 255 > cap) return 0;   // lz4.c:762
        out[op++] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
        if (run >= 15) op += lane_put_length(out + op, run - 15);
        lane_copy(out + op, in + anchor, run);
        op += run;
    }
    return op;
}

// Persistent grid: every lane pulls block indices from `counter` until the batch is exhausted.
// The slab -- kLaneTableBytes of hash table per lane of the grid -- is a set of equally sized CHUNKS, separate allocations: table g lives
// in chunk g / tables_per_chunk (a multiple of 64: one chunk per wavefront).  Why chunks: the rate of this kernel is the device's rate of
// random 64-byte read-modify-writes, and that rate depends on how the slab is spread over device memory -- 20.5 G per second inside one
// contiguous 8 GiB allocation, 25-27 G when it is spread (tools/microbench_random_sectors*.hip, profiles/r04/random_sectors_*.txt; the
// "two rates" of this encoder, 44-46 and 53-54 GB/s, were whether its one allocation happened to straddle such a boundary).  The host
// side builds a few candidate sets of chunks, measures each with slab_probe_kernel and keeps the fastest (launch_encode).
// only_deferred != 0: just the blocks the wavefront-per-block launch handed over (result[] == kDeferredResult).
__global__ void __launch_bounds__(64) encode_fast_lane_kernel(Batch b, unsigned long long* counter, uint8_t* const* chunks, unsigned tables_per_chunk, int only_deferred)
{
    const unsigned g = blockIdx.x * 64u + threadIdx.x;
    uint8_t* table = chunks[g / tables_per_chunk] + (size_t)(g % tables_per_chunk) * kLaneTableBytes;
    int epoch = 63;                                                   // the slab's contents are unknown at launch
    for (;;) {
        const int64_t blk = (int64_t)atomicAdd(counter, 1ull);
        if (blk >= b.n_blocks) return;
        if (only_deferred && b.result[blk] != kDeferredResult) continue;
        const int n = batch_src_len(b, blk), cap = batch_dst_cap(b, blk);
        const uint8_t* src = batch_src(b, blk);
        uint8_t* dst = batch_dst(b, blk);
        b.result[blk] = n < k64kLimit ? lane_encode_fast_block<false>(src, n, dst, cap, table, epoch)      // lz4.c:783-785
                                      : lane_encode_fast_block<true>(src, n, dst, cap, table, epoch);
    }
}

// What the slab's placement is worth: every lane of the encoder's grid performs `steps` DEPENDENT random 4-byte read-modify-writes inside its
// own table -- the encoder's memory behaviour without the encoder (one sector in, one sector out per step, nothing stays in a cache).  Timed by
// the host with events; leaves garbage in the tables (the encoder zeroes a table before its first use: epoch 63).
__global__ void __launch_bounds__(64) slab_probe_kernel(uint8_t* const* chunks, unsigned tables_per_chunk, int steps, unsigned* sink)
{
    const unsigned g = blockIdx.x * 64u + threadIdx.x;
    uint32_t* const p = (uint32_t*)(chunks[g / tables_per_chunk] + (size_t)(g % tables_per_chunk) * kLaneTableBytes);
    uint32_t s = g * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < steps; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t off = (s >> 4) % (uint32_t)(kLaneTableBytes / 4);
        const uint32_t v = p[off];
        p[off] = v + (uint32_t)i;
        s ^= v * 0x9E3779B9u;                                        // the next address depends on what was read
        acc += v;
    }
    if (acc == 0x12345678u) *sink = acc;
}

}  // namespace lz4hip
