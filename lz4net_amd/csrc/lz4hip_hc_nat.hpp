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

#ifdef LZ4HIP_TUNING_BUILD          /* the lane kernel lz4hip_hc_lcp.hpp replaced: kept for A/B runs (tools/hc_gen_ab.py) and the emulator tests */
// One lane = one block at a time (blocks [first, first + count), handed out by an atomic counter); all 64 lanes of the
// wavefront iterate together until every lane has run out of blocks.  The state machine of lz4hip_hc_conv.hpp without
// kHsZero / kHsInsert; chains + k * kHcNatChainBytes holds the natural chain of block first + k (hc_nat_chain_kernel).
// (100 VGPRs: four wavefronts per SIMD; a budget for five -- 96 and 16 bytes of scratch -- is slower, and 8 wavefronts per CU
//  are within 10 % of 16: the kernel sits at the fabric's random-sector rate, profiles/r03/hc_precomputed_chains.txt)
__global__ void __launch_bounds__(64) encode_hc_nat_kernel(Batch b, long long first, long long count, unsigned long long* counter, uint8_t* chains)
{
    // ---- the block ----
    const uint8_t* in = nullptr;
    uint8_t* out = nullptr;
    uint16_t* chain = nullptr;
    int64_t blk = 0;
    int n = 0, cap = 0, mflimit = 0, matchlimit = 0;
    // ---- the parse (variables of LZ4_compressHCCtx, lz4hc.c:553-742) ----
    int ip = 0, anchor = 0, op = 0;
    int ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0 = 0, ref0 = 0;
    int ml = 0, ml2 = 0, ml3 = 0, ml0 = 0;
    int phase = 0;                 // which search the control flow is waiting for: 0 best (ip), 1 wider -> ml2, 2 wider -> ml3
    // ---- the search in progress ----
    int st = kHsFetch;
    int s_ip = 0, s_limit = 0, s_back = 0;      // position searched, start limit (wider), ip - start_limit
    int s_len = 0;                 // best length so far (ml / longest)
    int s_match = 0, s_start = 0;  // where it was found (and, wider, where it starts)
    int s_ref = 0, s_link = 0, attempts = 0;
    uint32_t s_word = 0, s_probe = 0;
    int s_probe_ok = 0;            // s_probe is in[ip + ml] resp. in[start_limit + longest] for the current s_len
    Vec16 s_fwd = { { 0, 0, 0, 0 } };
    int s_fwd_ok = 0, s_repl = 0, s_delta = 0;
    // length counts (kHsFwd / kHsBack), and the walk of the repeat fill (c_s .. c_r)
    int c_n = 0, c_for_rep = 0, c_s = 0, c_r = 0, c_fwd_end = 0;

    // search request (from the control flow): LZ4HC_InsertAndFindBestMatch / LZ4HC_InsertAndGetWiderMatch
    auto request = [&](int pos, int start_limit, int longest, int match0, int start0_) {
        s_ip = pos; s_limit = start_limit; s_back = pos - start_limit; s_len = longest; s_match = match0; s_start = start0_;
        attempts = kHcAttempts; s_repl = 0; s_delta = 0; s_probe_ok = 0;
        st = kHsHead;
    };

    int it = 0;
    for (;;) {
        // ================= rare: block hand-out and the control flow between two searches =================
        if (st == kHsFetch) {
            const long long k = (long long)atomicAdd(counter, 1ull);
            if (k >= count) st = kHsExit;
            else {
                blk = (int64_t)first + k;
                chain = (uint16_t*)(chains + (size_t)k * kHcNatChainBytes);
                n = batch_src_len(b, blk); cap = batch_dst_cap(b, blk);
                in = batch_src(b, blk); out = batch_dst(b, blk);
                if (n > 65536) { b.result[blk] = -2000000002; st = kHsFetch; }   // LZ4HIP_E_ARGUMENT: this launch is for blocks <= 64 KiB
                else {
                    mflimit = n - kMfLimit; matchlimit = n - kLastLiterals;
                    ip = 1; anchor = 0; op = 0;                                  // lz4hc.c:581
                    phase = 3; st = kHsCtrl;                                     // (blocks too short for any match go straight to the last literals)
                }
            }
        }
        if (!wv::any(st != kHsExit)) break;

#include "lz4hip_hc_parse.inc"

        // ================= one memory step of the state each lane is in =================
        // Every load of the step is issued first -- seven load instructions, each carrying the lanes whose state needs it, at
        // per-lane addresses -- then the states are processed on what came back: ONE fabric round trip per iteration.
        const bool inH = st == kHsHead, inR = st == kHsRep, inP = st == kHsHop, inF = st == kHsFwd, inB = st == kHsBack, inL = st == kHsRepl;
        // Fwd: 16-byte pieces while they fit below matchlimit; Back: 4 bytes at a time while both sides have them
        const int f_a = s_ip + 4 + c_n, f_b = s_ref + 4 + c_n;
        const bool fwd16 = inF & (f_a + 16 <= matchlimit);
        const bool back4 = inB & (c_s - s_limit >= 4) & (c_r >= 4);
        // (1) chain entry: of the search position (ip - chain[ip] is its bucket's head) / of the candidate (next link)
        uint32_t v_link = 0;
        if (inH | inR | inP) v_link = chain[inH ? s_ip : s_ref];
        // (2) a word of the input: the search word, the candidate's word, 4 bytes before the start (backward extension)
        uint32_t v_w = 0;
        if (inH | inR | inP | back4) v_w = load_u32(in + (inH ? s_ip : inB ? c_s - 4 : s_ref));
        // (3) 4 bytes before the candidate's start (backward extension)
        uint32_t v_w2 = 0;
        if (back4) v_w2 = load_u32(in + c_r - 4);
        // (4) the candidate's byte at the best length so far, (5) the search position's byte there when the length has changed
        uint32_t v_cb = 0, v_pb = 0;
        if (inP) v_cb = phase == 0 ? in[s_ref + s_len] : in[s_ref - s_back + s_len];
        if (inP & (s_probe_ok == 0)) v_pb = phase == 0 ? in[s_ip + s_len] : in[s_limit + s_len];
        // (6) 16 bytes of the candidate's side (length count) / after the search position (search start), (7) the position's side
        Vec16 v_y = { { 0, 0, 0, 0 } }, v_x = { { 0, 0, 0, 0 } };
        {
            const bool fwd_ok_now = s_ip + 4 + 16 <= matchlimit;
            if (fwd16 | (inH & fwd_ok_now)) v_y = load_v16(in + (inF ? f_b : s_ip + 4));
            if (fwd16 & !((c_n == 0) & (s_fwd_ok != 0))) v_x = load_v16(in + f_a);
        }

        // ---- process ----
        if (inH) {                                                   // HASH_POINTER(ip) after LZ4HC_Insert(ip) == ip - natural chain[ip]
            s_word = v_w;
            s_fwd_ok = s_ip + 4 + 16 <= matchlimit;
            s_fwd = v_y;
            s_ref = s_ip - (int)v_link;
            if (phase == 0 && s_ref >= s_ip - 4) st = kHsRep;
            else st = (s_ref >= s_ip - kMaxDistance && s_ref >= 0) ? (int)kHsHop : (int)kHsCtrl;
        } else if (inR) {                                            // lz4hc.c:411-421
            s_link = (int)v_link;
            if (v_w == s_word) {
                s_delta = (s_ip - s_ref) & 0xFFFF;
                c_n = 0; c_for_rep = 1; st = kHsFwd;                 // repl = ml = common length + 4 (set when the count is complete)
            } else {
                s_ref -= s_link;
                st = (s_ref >= s_ip - kMaxDistance && s_ref >= 0) ? (int)kHsHop : (int)kHsCtrl;
            }
        } else if (inP) {                                            // lz4hc.c:424-434 / :481-516, one candidate
            attempts--;
            s_link = (int)v_link;
            if (!s_probe_ok) { s_probe = v_pb; s_probe_ok = 1; }     // *(ip + ml) resp. *(startLimit + longest): re-read only when the best length has changed
            if (v_cb == s_probe && v_w == s_word) {
                c_n = 0; c_for_rep = 0; st = kHsFwd;
            } else {
                s_ref -= s_link;
                if (!(s_ref >= s_ip - kMaxDistance && attempts > 0 && s_ref >= 0)) st = (s_repl && phase == 0) ? (int)kHsRepl : (int)kHsCtrl;
            }
        } else if (inF) {                                            // common length of in[s_ip + 4 + ..] and in[s_ref + 4 + ..] up to matchlimit
            int add = 0;
            bool more = false;
            if (fwd16) {
                const Vec16 x = ((c_n == 0) & (s_fwd_ok != 0)) ? s_fwd : v_x;
                const uint64_t d0 = (x.w[0] ^ v_y.w[0]) | ((uint64_t)(x.w[1] ^ v_y.w[1]) << 32);
                const uint64_t d1 = (x.w[2] ^ v_y.w[2]) | ((uint64_t)(x.w[3] ^ v_y.w[3]) << 32);
                if (d0) add = __builtin_ctzll(d0) >> 3;
                else if (d1) add = 8 + (__builtin_ctzll(d1) >> 3);
                else { add = 16; more = true; }
            } else {                                                 // the last bytes before matchlimit, one by one (rare)
                while (f_a + add < matchlimit && in[f_a + add] == in[f_b + add]) add++;
            }
            c_n += add;
            if (!more) {
                const int len = c_n + 4;
                if (c_for_rep) {                                     // lz4hc.c:416-418
                    s_repl = s_len = len; s_match = s_ref; s_probe_ok = 0;
                    s_ref -= s_link;
                    st = (s_ref >= s_ip - kMaxDistance && s_ref >= 0) ? (int)kHsHop : (int)kHsRepl;
                } else if (phase == 0) {                             // lz4hc.c:430-431
                    if (len > s_len) { s_len = len; s_match = s_ref; s_probe_ok = 0; }
                    s_ref -= s_link;
                    if (!(s_ref >= s_ip - kMaxDistance && attempts > 0 && s_ref >= 0)) st = s_repl ? (int)kHsRepl : (int)kHsCtrl;
                    else st = kHsHop;
                } else {                                             // wider: now backwards, lz4hc.c:505
                    c_fwd_end = s_ip + len; c_s = s_ip; c_r = s_ref; st = kHsBack;
                }
            }
        } else if (inB) {
            bool more = false;
            if (back4) {
                const uint32_t d = v_w ^ v_w2;                       // bytes c_s-4 .. c_s-1 against c_r-4 .. c_r-1: count from the top
                const int k = d == 0 ? 4 : (__builtin_clz(d) >> 3);
                c_s -= k; c_r -= k; more = k == 4;
            } else {
                for (int k = 0; k < 4; k++) {
                    if (c_s > s_limit && c_r > 0 && in[c_s - 1] == in[c_r - 1]) { c_s--; c_r--; more = k == 3; }
                    else { more = false; break; }
                }
            }
            if (!more) {                                             // lz4hc.c:507-512
                if (c_fwd_end - c_s > s_len) { s_len = c_fwd_end - c_s; s_match = c_r; s_start = c_s; s_probe_ok = 0; }
                s_ref -= s_link;
                st = (s_ref >= s_ip - kMaxDistance && attempts > 0 && s_ref >= 0) ? (int)kHsHop : (int)kHsCtrl;
            }
        } else if (inL) {                                            // lz4hc.c:437-455: DELTANEXT(q) = delta for q in [ip, end); a full
            if (s_repl > 0) { c_s = s_ip; c_r = s_ip + s_repl - 3; s_repl = -1; }   // group of 8 entries per step, else one entry
            int q = c_s;
            const uint32_t d = (uint32_t)s_delta;
            if ((q & 7) == 0 && q + 8 <= c_r) {
                const uint32_t dd = d | (d << 16);
                store_v16((uint8_t*)(chain + q), Vec16{ { dd, dd, dd, dd } });
                q += 8;
            } else {
                chain[q] = (uint16_t)d;
                q++;
            }
            c_s = q;
            if (q >= c_r) { s_repl = 0; st = kHsCtrl; }
        }
    }
}

#endif  // LZ4HIP_TUNING_BUILD

}  // namespace lz4hip
