// (NOT part of the product library: round 2's LZ4HC lane kernel, kept under tools/ab/ for A/B runs and emulator tests)
// lz4hip_hc_lane.hpp -- batched LZ4HC block encoder for gfx950, one LANE per block, bit-exact to
// the reference (same functions as lz4hip_hc.hpp: LZ4_compressHCCtx and its match finder,
// original/lz4hc.c:330-755 == src/LZ4pn/LZ4Codec.Unsafe64HC.Dirty.cs:72-523).
//
// LZ4HC is pointer chasing: ~110 k dependent chain hops and 65 k table inserts per 64 KiB block of
// fuzzer-style data.  One wavefront per block (lz4hip_hc.hpp) evaluates 64 candidates at once but
// still walks every chain serially at memory latency with 2 blocks per CU in flight.  Here every
// lane runs the whole algorithm for its own block, so a CU has a thousand chains in flight; heads
// and chain live in a per-lane global slab (192 KiB for blocks <= 64 KiB, 256 KiB above).  Persistent
// grid, work handed out per lane by an atomic counter.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_encode_lane.hpp"   // lane_count_equal, lane_copy, lane_put_length
#include "lz4hip_hc.hpp"            // hash15
#include "lz4hip_hc_conv.hpp"       // slab sizes

namespace lz4hip {

// (kHcLaneWavesPerCu, kHcLaneSlab16 / 32: lz4hip_hc_conv.hpp, which the product keeps)
#if 0
constexpr int kHcLaneWavesPerCu = 16;   // one block takes a lane ~2 s of dependent memory round trips: throughput = lanes in flight (4: 1.8, 8: 3.0, 16: 4.8, 20: 4.8 GB/s)
constexpr size_t kHcLaneSlab16 = 65536 + 131072;    // u16 heads + u16 chain
constexpr size_t kHcLaneSlab32 = 131072 + 131072;   // u32 heads + u16 chain
#endif

template <class HeadT>
struct LaneHc {
    HeadT* head;            // [32768], zero-filled per block (empty bucket == position 0, lz4hc.c:332)
    uint16_t* chain;        // [65536], slot = position & 0xFFFF; only slot 0 needs the 0xFFFF init (see lz4hip_hc.hpp)
    const uint8_t* in;
    int next;               // nextToUpdate

    // lz4hc.c:358-373
    LZ4HIP_DEVICE void insert_upto(int ip)
    {
        while (next < ip) {
            const int p = next;
            const uint32_t h = hash15(load_u32(in + p));
            const int prev = (int)head[h];
            const uint32_t delta = (p < prev || p - prev > kMaxDistance) ? (uint32_t)kMaxDistance : (uint32_t)(p - prev);
            chain[p & 0xFFFF] = (uint16_t)delta;
            head[h] = (HeadT)p;
            next++;
        }
    }

    // lz4hc.c:394-459
    // (The chain walk is what the kernel spends its time in and each lane-level memory request costs the CU's
    //  address unit a few cycles whatever it hits, so the walk makes as few as it can: the byte in[ip + ml] the
    //  reference re-reads for every candidate lives in a register and is re-read only when ml changes, the first
    //  16 bytes after in[ip + 4] are loaded once per search, and the next link is requested before the candidate is
    //  looked at.)
    LZ4HIP_DEVICE int best_match(int ip, int matchlimit, int& match_at)
    {
        int attempts = kHcAttempts, ml = 0, repl = 0, delta = 0;
        insert_upto(ip);
        const uint32_t word = load_u32(in + ip);
        const bool fwd_ok = ip + 4 + 16 <= matchlimit;
        const Vec16 fwd = fwd_ok ? load_v16(in + ip + 4) : Vec16{ { 0, 0, 0, 0 } };
        int ref = (int)head[hash15(word)];
        if (ref >= ip - 4) {                                           // lz4hc.c:411-421
            if (load_u32(in + ref) == word) {
                delta = (ip - ref) & 0xFFFF;
                repl = ml = lane_count_equal_from(in, ip + 4, ref + 4, matchlimit, fwd, fwd_ok) + 4;
                match_at = ref;
            }
            ref -= (int)chain[ref & 0xFFFF];
        }
        uint32_t probe_byte = in[ip + ml];                             // *(ip+ml), lz4hc.c:427
        while (ref >= ip - kMaxDistance && attempts > 0) {             // lz4hc.c:424-434
            attempts--;
            if (ref < 0) break;                                        // cannot happen on the reference's flows
            const int link = (int)chain[ref & 0xFFFF];
            if (in[ref + ml] == probe_byte && load_u32(in + ref) == word) {
                const int cand = lane_count_equal_from(in, ip + 4, ref + 4, matchlimit, fwd, fwd_ok) + 4;
                if (cand > ml) { ml = cand; match_at = ref; probe_byte = in[ip + ml]; }
            }
            ref -= link;
        }
        if (repl) {                                                    // lz4hc.c:437-455
            int q = ip;
            const int end = ip + repl - 3;
            while (q < end - delta) { chain[q & 0xFFFF] = (uint16_t)delta; q++; }
            do {
                chain[q & 0xFFFF] = (uint16_t)delta;
                head[hash15(load_u32(in + q))] = (HeadT)q;
                q++;
            } while (q < end);
            next = end;
        }
        return ml;
    }

    // lz4hc.c:462-518
    LZ4HIP_DEVICE int wider_match(int ip, int start_limit, int matchlimit, int longest, int& match_at, int& start_at)
    {
        int attempts = kHcAttempts;
        const int back = ip - start_limit;
        insert_upto(ip);
        const uint32_t word = load_u32(in + ip);
        const bool fwd_ok = ip + 4 + 16 <= matchlimit;
        const Vec16 fwd = fwd_ok ? load_v16(in + ip + 4) : Vec16{ { 0, 0, 0, 0 } };
        int ref = (int)head[hash15(word)];
        uint32_t probe_byte = in[start_limit + longest];               // *(startLimit + longest), lz4hc.c:480
        while (ref >= ip - kMaxDistance && attempts > 0) {
            attempts--;
            if (ref < 0) break;
            const int link = (int)chain[ref & 0xFFFF];
            if (in[ref - back + longest] == probe_byte && load_u32(in + ref) == word) {
                const int fwd_end = ip + 4 + lane_count_equal_from(in, ip + 4, ref + 4, matchlimit, fwd, fwd_ok);
                int s = ip, r = ref;
                while (s > start_limit && r > 0 && in[s - 1] == in[r - 1]) { s--; r--; }   // lz4hc.c:505
                if (fwd_end - s > longest) { longest = fwd_end - s; match_at = r; start_at = s; probe_byte = in[start_limit + longest]; }
            }
            ref -= link;
        }
        return longest;
    }
};

// lz4hc.c:521-550.  Returns false on output-limit hit.
LZ4HIP_DEVICE bool lane_hc_emit(const uint8_t* __restrict__ in, uint8_t* out, int& op, int cap, int& ip, int& anchor, int ml, int ref)
{
    const int ll = ip - anchor;
    const int token_at = op++;
    if (op + ll + 8 + (ll >> 8) > cap) return false;                   // lz4hc.c:529
    uint32_t token = ll >= 15 ? 0xF0u : (uint32_t)(ll << 4);
    if (ll >= 15) op += lane_put_length(out + op, ll - 15);
    lane_copy(out + op, in + anchor, ll);
    op += ll;
    const uint32_t off = (uint32_t)(ip - ref) & 0xFFFFu;
    out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8);
    op += 2;
    const int extra = ml - kMinMatch;
    if (op + 6 + (ll >> 8) > cap) return false;                        // lz4hc.c:541 tests the LITERAL length
    if (extra >= 15 && op + (extra - 15) / 255 + 1 > cap) return false;   // never write past cap (see lz4hip_hc.hpp)
    token |= extra >= 15 ? 15u : (uint32_t)extra;
    out[token_at] = (uint8_t)token;
    if (extra >= 15) op += lane_put_length(out + op, extra - 15);
    ip += ml;
    anchor = ip;
    return true;
}

template <class HeadT>
LZ4HIP_DEVICE int lane_encode_hc_block(const uint8_t* __restrict__ in, int n, uint8_t* out, int cap, uint8_t* slab)
{
    LaneHc<HeadT> st;
    st.head = (HeadT*)slab;
    st.chain = (uint16_t*)(slab + 32768 * sizeof(HeadT));
    st.in = in; st.next = 1;                                           // lz4hc.c:334
    for (int k = 0; k < (int)(32768 * sizeof(HeadT)); k += 16) store_v16(slab + k, Vec16{ { 0, 0, 0, 0 } });
    st.chain[0] = 0xFFFF;

    const int mflimit = n - kMfLimit, matchlimit = n - kLastLiterals;
    int ip = 0, anchor = 0, op = 0;
    int ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0, ref0;
    int ml, ml2, ml3, ml0;

    ip++;
    while (ip < mflimit) {                                             // lz4hc.c:584
        ml = st.best_match(ip, matchlimit, ref);
        if (!ml) { ip++; continue; }
        start0 = ip; ref0 = ref; ml0 = ml;
        bool search3 = false;
        for (;;) {
            if (!search3) {                                            // _Search2, lz4hc.c:594-622
                ml2 = (ip + ml < mflimit) ? st.wider_match(ip + ml - 2, ip + 1, matchlimit, ml, ref2, start2) : ml;
                if (ml2 == ml) { if (!lane_hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0; break; }
                if (start0 < ip && start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
                if (start2 - ip < 3) { ml = ml2; ip = start2; ref = ref2; continue; }
            }
            search3 = true;                                            // _Search3, lz4hc.c:624-727
            if (start2 - ip < kHcOptimalMl) {
                int new_ml = ml > kHcOptimalMl ? kHcOptimalMl : ml;
                if (ip + new_ml > start2 + ml2 - kMinMatch) new_ml = (start2 - ip) + ml2 - kMinMatch;
                const int corr = new_ml - (start2 - ip);
                if (corr > 0) { start2 += corr; ref2 += corr; ml2 -= corr; }
            }
            ml3 = (start2 + ml2 < mflimit) ? st.wider_match(start2 + ml2 - 3, start2, matchlimit, ml2, ref3, start3) : ml2;
            if (ml3 == ml2) {
                if (start2 < ip + ml) ml = start2 - ip;
                if (!lane_hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
                ip = start2;
                if (!lane_hc_emit(in, out, op, cap, ip, anchor, ml2, ref2)) return 0;
                break;
            }
            if (start3 < ip + ml + 3) {
                if (start3 >= ip + ml) {
                    if (start2 < ip + ml) {
                        const int corr = ip + ml - start2;
                        start2 += corr; ref2 += corr; ml2 -= corr;
                        if (ml2 < kMinMatch) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                    }
                    if (!lane_hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
                    ip = start3; ref = ref3; ml = ml3;
                    start0 = start2; ref0 = ref2; ml0 = ml2;
                    search3 = false;
                    continue;
                }
                start2 = start3; ref2 = ref3; ml2 = ml3;
                continue;
            }
            if (start2 < ip + ml) {
                if (start2 - ip < 15) {
                    if (ml > kHcOptimalMl) ml = kHcOptimalMl;
                    if (ip + ml > start2 + ml2 - kMinMatch) ml = (start2 - ip) + ml2 - kMinMatch;
                    const int corr = ml - (start2 - ip);
                    if (corr > 0) { start2 += corr; ref2 += corr; ml2 -= corr; }
                } else {
                    ml = start2 - ip;
                }
            }
            if (!lane_hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
            ip = start2; ref = ref2; ml = ml2;
            start2 = start3; ref2 = ref3; ml2 = ml3;
        }
    }
    {   // last literals, lz4hc.c:730-738
        const int run = n - anchor;
        if (op + run + 1 + (run + 255 - 15) / 255 > cap) return 0;
        out[op++] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
        if (run >= 15) op += lane_put_length(out + op, run - 15);
        lane_copy(out + op, in + anchor, run);
        op += run;
    }
    return op;
}

#ifdef LZ4HIP_TUNING_BUILD          /* round 2's kernel: launched only by tuning builds (tools/hc_gen_ab.py) and the emulator tests */
// Persistent grid; `slabs` holds slab_bytes per lane of the grid.
__global__ void __launch_bounds__(64) encode_hc_lane_kernel(Batch b, unsigned long long* counter, uint8_t* slabs, unsigned long long slab_bytes)
{
    uint8_t* slab = slabs + ((size_t)blockIdx.x * 64 + threadIdx.x) * (size_t)slab_bytes;
    for (;;) {
        const int64_t blk = (int64_t)atomicAdd(counter, 1ull);
        if (blk >= b.n_blocks) return;
        const int n = batch_src_len(b, blk), cap = batch_dst_cap(b, blk);
        const uint8_t* src = batch_src(b, blk);
        uint8_t* dst = batch_dst(b, blk);
        int r;
        if (n <= 65536)                       r = lane_encode_hc_block<uint16_t>(src, n, dst, cap, slab);
        else if (slab_bytes >= kHcLaneSlab32) r = lane_encode_hc_block<uint32_t>(src, n, dst, cap, slab);
        else                                  r = -2000000002;   // LZ4HIP_E_ARGUMENT: launch reserved 16-bit heads only
        b.result[blk] = r;
    }
}
#endif  // LZ4HIP_TUNING_BUILD

}  // namespace lz4hip
