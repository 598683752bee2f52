// lz4hip_hc.hpp -- batched LZ4HC block encoder for gfx950, one wavefront per block, bit-exact to
// the reference.
//
// Replaces LZ4_compressHC_limitedOutput -> LZ4_compressHCCtx (original/lz4hc.c:745-755,557-742 ==
// LZ4_compressHCCtx_64, src/LZ4pn/LZ4Codec.Unsafe64HC.Dirty.cs:331-523) with its match finder
// LZ4HC_Insert / InsertAndFindBestMatch / InsertAndGetWiderMatch (original/lz4hc.c:358-518 ==
// Unsafe64HC.Dirty.cs:72-263) and LZ4_encodeSequence (lz4hc.c:521-550 == :265-329).
//
// What is parallel here (and provably equal to the serial reference):
//  * chain insertion (lz4hc.c:358-373) of up to 64 consecutive positions per step: every lane
//    hashes its own position; same-bucket positions inside the step are chained to each other in
//    position order (the nearest lower lane with the same hash), the first of each bucket chains to
//    the head read before the step, the last of each bucket becomes the new head;
//  * candidate evaluation: the chain is walked on scalars (it is a dependent pointer chase), up to
//    64 candidates are collected, then every lane measures one candidate (forward common length,
//    and for the "wider" search the backward extension).  The reference keeps a candidate only if it
//    is STRICTLY longer than the best so far and its 1-byte pre-check is a pure filter, so its result
//    is "the first candidate in chain order that attains the maximum" == lowest lane with the maximum;
//  * literal copies / length fills as in the fast encoder.
// The lazy 3-match parser (lz4hc.c:584-727) is control flow on wave-uniform scalars.
//
// State per block: heads u16[32768] (blocks <= 64 KiB; u32 for larger ones) in LDS, zero-filled
// (empty bucket == position 0, lz4hc.c:332); chain u16[65536] in a per-workgroup slab of global
// memory (L2 resident).  Only slot 0 of the chain needs the reference's 0xFFFF initialisation: every
// other slot is written when its position is inserted, before any walk can reach it.
// The grid is persistent (kHcGroupsPerCu workgroups per CU, work handed out by an atomic counter)
// so that the chain slabs are bounded by residency, not by the batch size.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_encode.hpp"   // input_word, wave_common_length, put_length_bytes

namespace lz4hip {

constexpr int kHcGroupsPerCu = 2;
constexpr int kHcGlobalBytesPerGroup = 131072;      // chain: u16[65536]
constexpr int kHcLdsHeads16 = 65536;                // u16[32768]
constexpr int kHcLdsHeads32 = 131072;               // u32[32768]
constexpr int kHcLdsBytes = kHcLdsHeads16;          // default launch (all blocks <= 64 KiB)

LZ4HIP_DEVICE uint32_t hash15(uint32_t word) { return (word * kGolden) >> 17; }   // lz4hc.c:180,245

// per-lane (divergent) exact common length of in[a..] and in[b..], a bounded by limit
LZ4HIP_DEVICE int lane_common_length(const uint8_t* in, int a, int b, int limit)
{
    int n = 0;
    while (a + n + 4 <= limit && load_u32(in + a + n) == load_u32(in + b + n)) n += 4;
    while (a + n < limit && in[a + n] == in[b + n]) n++;
    return n;
}

LZ4HIP_DEVICE uint32_t wave_max_u32(uint32_t v)
{
    const int lane = wv::lane();
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = wv::shuffle(v, lane ^ d);
        v = o > v ? o : v;
    }
    return v;
}

template <class HeadT>
struct HcState {
    HeadT* head;            // LDS
    uint16_t* chain;        // global slab of this workgroup
    const uint8_t* in;
    int next;               // nextToUpdate (lz4hc.c:236)

    LZ4HIP_DEVICE int chain_at(int pos) const { return (int)wv::uniform((uint32_t)chain[pos & 0xFFFF]); }

    // lz4hc.c:358-373, 64 positions per step
    LZ4HIP_DEVICE void insert_upto(int ip)
    {
        const int lane = wv::lane();
        while (next < ip) {
            const int cnt = ip - next < 64 ? ip - next : 64;
            const bool active = lane < cnt;
            const int p = next + lane;
            const uint32_t h = active ? hash15(load_u32(in + p)) : 0xFFFFFFFFu;
            // lanes of this step that share my bucket
            uint64_t same = 0;
            for (int k = 0; k < cnt; k++) {
                const uint32_t hk = wv::readlane(h, k);
                const uint64_t m = wv::ballot(active && h == hk);
                if (lane == k) same = m;
            }
            const uint64_t below = same & ((1ull << lane) - 1ull);
            int prev = 0;
            if (active) prev = below ? next + (63 - __builtin_clzll(below)) : (int)head[h];
            wv::mem_sync();                                   // all head reads before any head write
            if (active) {
                // size_t delta = p - HASH_POINTER(p); if (delta > MAX_DISTANCE) delta = MAX_DISTANCE;
                const uint32_t delta = (p < prev || p - prev > kMaxDistance) ? (uint32_t)kMaxDistance : (uint32_t)(p - prev);
                chain[p & 0xFFFF] = (uint16_t)delta;
                if ((same >> lane) >> 1 == 0) head[h] = (HeadT)p;   // last position of this bucket in the step
            }
            wv::mem_sync();
            next += cnt;
        }
    }

    // Walk the chain from `ref` collecting up to 64 candidates (lane i keeps the i-th).
    // Returns the number collected; updates ref / attempts like the reference's loop header
    // `while ((ref >= ip-MAX_DISTANCE) && (nbAttempts))` (lz4hc.c:424,474).
    LZ4HIP_DEVICE int collect(int ip, int& ref, int& attempts, int& cand) const
    {
        const int lane = wv::lane();
        int cnt = 0;
        cand = 0;
        while (cnt < 64 && ref >= ip - kMaxDistance && attempts > 0) {
            if (ref < 0) { attempts = 0; break; }             // cannot happen on the reference's flows
            attempts--;
            if (lane == cnt) cand = ref;
            cnt++;
            ref -= chain_at(ref);
        }
        return cnt;
    }

    // lz4hc.c:394-459
    LZ4HIP_DEVICE int best_match(int ip, int matchlimit, int& match_at)
    {
        const int lane = wv::lane();
        int attempts = kHcAttempts, ml = 0, repl = 0, delta = 0;
        insert_upto(ip);
        const uint32_t word = input_word(in, ip);
        int ref = (int)wv::uniform((uint32_t)head[hash15(word)]);

        if (ref >= ip - 4) {                                           // lz4hc.c:411-421
            if (input_word(in, ref) == word) {
                delta = (ip - ref) & 0xFFFF;
                repl = ml = wave_common_length(in, ip + 4, ref + 4, matchlimit) + 4;
                match_at = ref;
            }
            ref -= chain_at(ref);
        }
        while (ref >= ip - kMaxDistance && attempts > 0) {             // lz4hc.c:424-434
            int cand;
            const int cnt = collect(ip, ref, attempts, cand);
            uint32_t len = 0;
            if (lane < cnt && load_u32(in + cand) == word)
                len = 4u + (uint32_t)lane_common_length(in, ip + 4, cand + 4, matchlimit);
            const uint32_t best = wave_max_u32(len);
            if ((int)best > ml) {
                const int w = wv::ctz64(wv::ballot(len == best));
                ml = (int)best;
                match_at = (int)wv::readlane((uint32_t)cand, w);
            }
        }
        if (repl) {                                                    // lz4hc.c:437-455
            const int end = ip + repl - 3;
            // while (ptr < end-delta) DELTANEXT(ptr++) = delta;
            const int pre_end = end - delta;
            for (int q = ip + lane; q < pre_end; q += 64) chain[q & 0xFFFF] = (uint16_t)delta;
            wv::mem_sync();
            int q = pre_end > ip ? pre_end : ip;
            do {                                                       // at most `delta` (<= 4) iterations beyond the pre-fill
                const uint32_t hq = hash15(input_word(in, q));
                if (lane == 0) { chain[q & 0xFFFF] = (uint16_t)delta; head[hq] = (HeadT)q; }
                q++;
            } while (q < end);
            wv::mem_sync();
            next = end;
        }
        return ml;
    }

    // lz4hc.c:462-518
    LZ4HIP_DEVICE int wider_match(int ip, int start_limit, int matchlimit, int longest, int& match_at, int& start_at)
    {
        const int lane = wv::lane();
        int attempts = kHcAttempts;
        insert_upto(ip);
        const uint32_t word = input_word(in, ip);
        int ref = (int)wv::uniform((uint32_t)head[hash15(word)]);
        while (ref >= ip - kMaxDistance && attempts > 0) {
            int cand;
            const int cnt = collect(ip, ref, attempts, cand);
            uint32_t total = 0;
            int back = 0;
            if (lane < cnt && load_u32(in + cand) == word) {
                const int fwd = 4 + lane_common_length(in, ip + 4, cand + 4, matchlimit);
                while (ip - back > start_limit && cand - back > 0 && in[ip - back - 1] == in[cand - back - 1]) back++;   // lz4hc.c:505
                total = (uint32_t)(fwd + back);
            }
            const uint32_t best = wave_max_u32(total);
            if ((int)best > longest) {
                const int w = wv::ctz64(wv::ballot(total == best));
                longest = (int)best;
                match_at = (int)wv::readlane((uint32_t)(cand - back), w);
                start_at = ip - (int)wv::readlane((uint32_t)back, w);
            }
        }
        return longest;
    }
};

// lz4hc.c:521-550.  Returns false on output-limit hit.
LZ4HIP_DEVICE bool hc_emit(const uint8_t* in, uint8_t* out, int& op, int cap, int& ip, int& anchor, int ml, int ref)
{
    const int lane = wv::lane();
    const int ll = ip - anchor;
    const int token_at = op++;
    if (op + ll + 8 + (ll >> 8) > cap) return false;                   // lz4hc.c:529
    uint32_t token = ll >= 15 ? 0xF0u : (uint32_t)(ll << 4);
    if (ll >= 15) op += put_length_bytes(out + op, ll - 15);
    wave_copy(out + op, in + anchor, ll);
    op += ll;
    const uint32_t off = (uint32_t)(ip - ref) & 0xFFFFu;
    if (lane == 0) { out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8); }
    op += 2;
    const int extra = ml - kMinMatch;
    if (op + 6 + (ll >> 8) > cap) return false;                        // lz4hc.c:541 tests the LITERAL length
    // The reference can run past `cap` right here and only then return 0 from a later check; the
    // result is the same 0, but this implementation never writes outside [out, out + cap).
    if (extra >= 15 && op + (extra - 15) / 255 + 1 > cap) return false;
    token |= extra >= 15 ? 15u : (uint32_t)extra;
    if (lane == 0) out[token_at] = (uint8_t)token;
    if (extra >= 15) op += put_length_bytes(out + op, extra - 15);
    ip += ml;
    anchor = ip;
    return true;
}

template <class HeadT>
LZ4HIP_DEVICE int encode_hc_block(const uint8_t* in, int n, uint8_t* out, int cap, unsigned char* lds, uint16_t* chain)
{
    const int lane = wv::lane();
    HcState<HeadT> st;
    st.head = (HeadT*)lds; st.chain = chain; st.in = in; st.next = 1;  // lz4hc.c:334 (LZ4_ARCH64: base + 1)
    for (int k = lane * 16; k < (int)(32768 * sizeof(HeadT)); k += 1024) *(Vec16*)(lds + k) = Vec16{ { 0, 0, 0, 0 } };
    if (lane == 0) chain[0] = 0xFFFF;                                  // lz4hc.c:333 (see header comment)
    wv::mem_sync();

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
                if (ml2 == ml) { if (!hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0; break; }
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
                if (!hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
                ip = start2;
                if (!hc_emit(in, out, op, cap, ip, anchor, ml2, ref2)) return 0;
                break;
            }
            if (start3 < ip + ml + 3) {
                if (start3 >= ip + ml) {
                    if (start2 < ip + ml) {
                        const int corr = ip + ml - start2;
                        start2 += corr; ref2 += corr; ml2 -= corr;
                        if (ml2 < kMinMatch) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                    }
                    if (!hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
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
            if (!hc_emit(in, out, op, cap, ip, anchor, ml, ref)) return 0;
            ip = start2; ref = ref2; ml = ml2;
            start2 = start3; ref2 = ref3; ml2 = ml3;
        }
    }
    {   // last literals, lz4hc.c:730-738
        const int run = n - anchor;
        if (op + run + 1 + (run + 255 - 15) / 255 > cap) return 0;
        if (lane == 0) out[op] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
        op++;
        if (run >= 15) op += put_length_bytes(out + op, run - 15);
        wave_copy(out + op, in + anchor, run);
        op += run;
    }
    return op;
}

// Persistent grid: `counter` hands out block indices; `chains` holds one 128 KiB chain slab per
// workgroup.  lds_bytes tells the kernel which head width the launch reserved.
__global__ void __launch_bounds__(64) encode_hc_kernel(Batch b, unsigned long long* counter, uint8_t* chains, int lds_bytes)
{
    LZ4HIP_DYN_LDS(lds);
    uint16_t* chain = (uint16_t*)(chains + (size_t)blockIdx.x * kHcGlobalBytesPerGroup);
    for (;;) {
        unsigned long long mine = 0;
        if (wv::lane() == 0) mine = atomicAdd(counter, 1ull);
        const int64_t blk = (int64_t)wv::first_lane((uint64_t)mine);
        if (blk >= b.n_blocks) return;
        const int n = wv::uniform(batch_src_len(b, blk));
        const int cap = wv::uniform(batch_dst_cap(b, blk));
        const uint8_t* src = batch_src(b, blk);
        uint8_t* dst = batch_dst(b, blk);
        int r;
        if (n <= 65536)                      r = encode_hc_block<uint16_t>(src, n, dst, cap, lds, chain);
        else if (lds_bytes >= kHcLdsHeads32) r = encode_hc_block<uint32_t>(src, n, dst, cap, lds, chain);
        else                                 r = -2000000002;   // LZ4HIP_E_ARGUMENT: launch reserved 16-bit heads only
        if (wv::lane() == 0) b.result[blk] = r;
        wv::mem_sync();
    }
}

}  // namespace lz4hip
