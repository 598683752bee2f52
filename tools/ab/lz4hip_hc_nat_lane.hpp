// lz4hip_hc_nat_lane.hpp -- NOT part of the product library: the LZ4HC lane kernel of round 3's second step (natural chains built up
// front by hc_nat_chain_kernel, lz4net_amd/csrc/lz4hip_hc_nat.hpp, walked WITHOUT shared lengths).  lz4hip_hc_lcp.hpp replaced it;
// it is kept for A/B runs (tools/hc_gen_ab.py, libraries built with -DLZ4HIP_TUNING_BUILD) and for the emulator tests of the
// chain builder's uint16_t variant.
#pragma once
#include "lz4hip_hc_nat.hpp"

namespace lz4hip {

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


}  // namespace lz4hip
