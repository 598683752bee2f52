// lz4hip_hc_conv.hpp -- batched LZ4HC block encoder for gfx950, one LANE per block like lz4hip_hc_lane.hpp and bit-exact
// to the same reference functions (LZ4_compressHCCtx and its match finder, original/lz4hc.c:330-755), but CONVERGENT.
//
// What the round-2 counters said about lz4hip_hc_lane.hpp (profiles/r02/pmc_encode_hc_lane_*.json): a wavefront issues
// 9.1 M load instructions for its 64 blocks although one block needs only ~0.4 M loads (110 k chain hops, 65 k inserts,
// counted with the instrumented oracle) -- on average fewer than 3 of the 64 lanes take part in a load.  The lanes run
// the same algorithm but sit in different loops of it (insert loop, chain walk of the best-match search, chain walk of
// the wider-match search, length count, sequence emit), the compiler serialises those loops, and every load of every
// loop is a dependent fabric round trip: 89 % of the wave-cycles are spent at s_waitcnt with the fabric at half of
// what it can serve (1.7 TB/s of sectors).
//
// Here the algorithm is a STATE MACHINE per lane and the kernel ONE loop: in every iteration every lane performs one
// memory step of whatever it is doing -- insert one position, follow one chain link, compare one 16-byte piece, extend one
// step backwards, fill one entry of the repeat optimisation -- and the light control flow between two searches (the lazy
// three-match parse, lz4hc.c:584-727, and the sequence emit) runs when a lane's search is complete.  Lanes in the same
// state execute that state's loads together, whichever search of whichever sequence they belong to, so a load
// instruction carries tens of lanes instead of three and a wavefront needs a few hundred thousand iterations, not nine
// million loads.
//
// State per lane in global memory exactly as in lz4hip_hc_lane.hpp: heads (zero-filled per block; empty bucket ==
// position 0, lz4hc.c:332) and the u16 chain (slot = position & 0xFFFF; only slot 0 needs the 0xFFFF init) in a per-lane
// slab.  Persistent grid, blocks handed out per lane by an atomic counter.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_encode_lane.hpp"   // lane_copy, lane_put_length
#include "lz4hip_hc.hpp"            // hash15

namespace lz4hip {

constexpr int kHcLaneWavesPerCu = 16;   // one block takes a lane ~2 s of dependent memory round trips: throughput = lanes in flight (4: 1.8, 8: 3.0, 16: 4.8, 20: 4.8 GB/s)
constexpr size_t kHcLaneSlab16 = 65536 + 131072;    // u16 heads + u16 chain
constexpr size_t kHcLaneSlab32 = 131072 + 131072;   // u32 heads + u16 chain

#ifndef LZ4HIP_HC_CTRL_EVERY
#define LZ4HIP_HC_CTRL_EVERY 8      /* 2 / 4 / 8 / 16: 7.9 / 8.1 / 8.3 / 8.3 GB/s on D2 (profiles/r03/hc_convergent_control_batching_interval.txt) */
#endif
constexpr int kHcCtrlEvery = LZ4HIP_HC_CTRL_EVERY;      // (power of two)
constexpr int kHcCtrlBatchLanes = 16;

enum HcConvState {
    kHsFetch = 0,   // take the next block from the counter
    kHsZero,        // zero the heads, 64 bytes per step
    kHsInsert,      // LZ4HC_Insert: one position per step, up to the search position
    kHsHead,        // the search position's word, the 16 bytes after it, its bucket's head
    kHsRep,         // best match only: the candidate within 4 bytes (repeat detection, lz4hc.c:411-421)
    kHsHop,         // one chain link: next link + the two probes of the candidate
    kHsFwd,         // forward length count, 16 bytes per step
    kHsBack,        // wider match only: backward extension, up to 4 bytes per step
    kHsRepl,        // best match only: repeat optimisation fill (lz4hc.c:437-455), one entry per step
    kHsCtrl,        // search complete: control flow up to the next search (or the end of the block)
    kHsExit
};

// One lane = one block at a time; all 64 lanes of the wavefront iterate together until every lane has run out of blocks.
// (113 VGPRs: four wavefronts per SIMD.  The rate still grows with the residency at 16 wavefronts per CU, but a register budget
//  for 5 / 6 per SIMD costs 18 / 36 dwords of scratch per lane and is slower: profiles/r03/hc_convergent_more_wavefronts_with_spills.txt)
template <class HeadT>
__global__ void __launch_bounds__(64) encode_hc_conv_kernel(Batch b, unsigned long long* counter, uint8_t* slabs, unsigned long long slab_bytes)
{
    uint8_t* const slab = slabs + ((size_t)blockIdx.x * 64 + threadIdx.x) * (size_t)slab_bytes;
    HeadT* const head = (HeadT*)slab;
    uint16_t* const chain = (uint16_t*)(slab + 32768 * sizeof(HeadT));

    // ---- the block ----
    const uint8_t* in = nullptr;
    uint8_t* out = nullptr;
    int64_t blk = 0;
    int n = 0, cap = 0, mflimit = 0, matchlimit = 0;
    // ---- the parse (variables of LZ4_compressHCCtx, lz4hc.c:553-742) ----
    int ip = 0, anchor = 0, op = 0, next = 1;
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
    int s_wok = 0;                 // s_word holds the word at s_ip
    int s_probe_ok = 0;            // s_probe is in[ip + ml] resp. in[start_limit + longest] for the current s_len
    Vec16 s_fwd = { { 0, 0, 0, 0 } };
    int s_fwd_ok = 0, s_repl = 0, s_delta = 0;
    int zero_at = 0;
    // length counts (kHsFwd / kHsBack)
    int c_n = 0, c_for_rep = 0, c_s = 0, c_r = 0, c_fwd_end = 0;
    // The insert loop reads the input through a 16-byte register window (bytes [iw_pos, iw_pos + iw_len) of the block): one
    // 16-byte load per 13 positions instead of a 4-byte load per position -- with a quarter of a million lanes in flight a
    // line does not survive in any cache between two steps of a lane, so every load is a line across the fabric.
    Vec16 iw = { { 0, 0, 0, 0 } };
    int iw_pos = 0, iw_len = 0;
    // ... and writes the chain through an 8-entry buffer (one aligned 16-byte store per group of 8 positions and per insert
    // burst instead of a 2-byte store per position).  Only for blocks <= 64 KiB (16-bit heads): the buffer also stores the
    // group's not-yet-inserted slots, which nobody reads before they are written -- unless slots wrap (position & 0xFFFF).
    constexpr bool kChainBuf = sizeof(HeadT) == 2;
    uint32_t cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0;
    int cb_base = 0, cb_dirty = 0;
    auto iw_has = [&](int p) { return (p >= iw_pos) & (p + 4 <= iw_pos + iw_len); };
    auto iw_word = [&](int p) -> uint32_t {                           // (iw_has(p))
        const int o = p - iw_pos, q = o >> 2;
        const uint32_t lo = q < 2 ? (q == 0 ? iw.w[0] : iw.w[1]) : (q == 2 ? iw.w[2] : iw.w[3]);
        const uint32_t hi = q < 2 ? (q == 0 ? iw.w[1] : iw.w[2]) : iw.w[3];
        return wv::alignbyte(hi, lo, (uint32_t)o & 3u);
    };
    auto chain_flush = [&]() {
        if (kChainBuf && cb_dirty) { store_v16((uint8_t*)(chain + cb_base), Vec16{ { cb0, cb1, cb2, cb3 } }); cb_dirty = 0; }
    };
    auto chain_put = [&](int p, uint32_t delta) {
        if (!kChainBuf) { chain[p & 0xFFFF] = (uint16_t)delta; return; }
        if ((p & ~7) != cb_base) { chain_flush(); cb_base = p & ~7; }
        const int k = (p & 7) >> 1;
        const uint32_t keep = (p & 1) ? 0x0000FFFFu : 0xFFFF0000u, val = (p & 1) ? (delta << 16) : delta;
        cb0 = k == 0 ? (cb0 & keep) | val : cb0; cb1 = k == 1 ? (cb1 & keep) | val : cb1;
        cb2 = k == 2 ? (cb2 & keep) | val : cb2; cb3 = k == 3 ? (cb3 & keep) | val : cb3;
        cb_dirty = 1;
    };

    // search request (from the control flow): LZ4HC_InsertAndFindBestMatch / LZ4HC_InsertAndGetWiderMatch
    auto request = [&](int pos, int start_limit, int longest, int match0, int start0_) {
        s_ip = pos; s_limit = start_limit; s_back = pos - start_limit; s_len = longest; s_match = match0; s_start = start0_;
        attempts = kHcAttempts; s_repl = 0; s_delta = 0; s_probe_ok = 0; s_wok = 0;
        st = next < pos ? (int)kHsInsert : (int)kHsHead;
    };

    int it = 0;
    for (;;) {
        // ================= rare: block hand-out and the control flow between two searches =================
        if (st == kHsFetch) {
            blk = (int64_t)atomicAdd(counter, 1ull);
            if (blk >= b.n_blocks) st = kHsExit;
            else {
                n = batch_src_len(b, blk); cap = batch_dst_cap(b, blk);
                in = batch_src(b, blk); out = batch_dst(b, blk);
                if (sizeof(HeadT) == 2 && n > 65536) { b.result[blk] = -2000000002; st = kHsFetch; }   // LZ4HIP_E_ARGUMENT: launch reserved 16-bit heads only
                else {
                    mflimit = n - kMfLimit; matchlimit = n - kLastLiterals;
                    ip = 1; anchor = 0; op = 0; next = 1; iw_len = 0;                    // lz4hc.c:334, :581
                    cb0 = 0xFFFFu; cb1 = cb2 = cb3 = 0; cb_base = 0; cb_dirty = 0;       // (slot 0 keeps its 0xFFFF through the flushes of group 0)
                    zero_at = 0; st = kHsZero;
                }
            }
        }
        if (!wv::any(st != kHsExit)) break;

#include "lz4hip_hc_parse.inc"

        // ================= one memory step of the state each lane is in =================
        // Every load of the step is issued first -- eight load instructions, each carrying the lanes whose state needs it, at
        // per-lane addresses -- then the states are processed on what came back: ONE fabric round trip per iteration whatever
        // mix of states the wavefront is in (a state-by-state body made the round trips of the states add up).
        if (st == kHsHead && !s_wok && iw_has(s_ip)) { s_word = iw_word(s_ip); s_wok = 1; }   // (the insert loop's window usually holds the search word)
        const bool inI = st == kHsInsert, inH = st == kHsHead, inR = st == kHsRep, inP = st == kHsHop, inF = st == kHsFwd,
                   inB = st == kHsBack, inL = st == kHsRepl;
        const bool ins_go = inI & iw_has(next), head_go = inH & (s_wok != 0);
        // the window has to move when the position after this one is not in it any more (or this one is not: then this step only loads)
        const int iw_want = ins_go ? next + 1 : next;
        const bool iw_load = inI & (!ins_go | ((next + 1 <= s_ip) & !iw_has(next + 1)));
        const bool iw_full = iw_want + 16 <= n;                      // (near the end of a block: 4 bytes at a time)
        const uint32_t ins_word = ins_go ? iw_word(next) : 0u;
        // Fwd: 16-byte pieces while they fit below matchlimit; Back: 4 bytes at a time while both sides have them
        const int f_a = s_ip + 4 + c_n, f_b = s_ref + 4 + c_n;
        const bool fwd16 = inF & (f_a + 16 <= matchlimit);
        const bool back4 = inB & (c_s - s_limit >= 4) & (c_r >= 4);
        const bool repl_head = inL & (s_repl < 0) & !(c_s < c_r - s_delta);      // (s_repl < 0: the walk is set up, see below)
        // (1) head of a bucket: insert / search start
        const uint32_t hsh = hash15(ins_go ? ins_word : s_word);
        HeadT v_head = 0;
        if (ins_go | head_go) v_head = head[hsh];
        // (2) chain link of the candidate
        uint32_t v_link = 0;
        if (inR | inP) v_link = chain[s_ref & 0xFFFF];
        // (3) a word of the input: the next position's word (insert), the search word, the candidate's word, 4 bytes before the
        //     start (backward extension), the word to hash (repeat fill)
        uint32_t v_w = 0;
        {
            const int aw = inI ? iw_want : inH ? s_ip : (inR | inP) ? s_ref : inB ? c_s - 4 : c_s;
            const bool need = (iw_load & !iw_full) | (inH & !head_go) | inR | inP | back4 | repl_head;
            if (need) v_w = load_u32(in + aw);
        }
        // (4) 4 bytes before the candidate's start (backward extension)
        uint32_t v_w2 = 0;
        if (back4) v_w2 = load_u32(in + c_r - 4);
        // (5) the candidate's byte at the best length so far, (6) the search position's byte there when the length has changed
        uint32_t v_cb = 0, v_pb = 0;
        if (inP) v_cb = phase == 0 ? in[s_ref + s_len] : in[s_ref - s_back + s_len];
        if (inP & (s_probe_ok == 0)) v_pb = phase == 0 ? in[s_ip + s_len] : in[s_limit + s_len];
        // (7) 16 bytes of the candidate's side (length count) / after the search position (search start), (8) the position's side
        Vec16 v_y = { { 0, 0, 0, 0 } }, v_x = { { 0, 0, 0, 0 } };
        {
            const bool fwd_ok_now = s_ip + 4 + 16 <= matchlimit;
            if (fwd16 | (head_go & fwd_ok_now) | (iw_load & iw_full)) v_y = load_v16(in + (inF ? f_b : inI ? iw_want : s_ip + 4));
            if (fwd16 & !((c_n == 0) & (s_fwd_ok != 0))) v_x = load_v16(in + f_a);
        }

        // ---- process ----
        if (st == kHsZero) {
            uint8_t* z = slab + zero_at;
            store_v16(z, Vec16{ { 0, 0, 0, 0 } }); store_v16(z + 16, Vec16{ { 0, 0, 0, 0 } });
            store_v16(z + 32, Vec16{ { 0, 0, 0, 0 } }); store_v16(z + 48, Vec16{ { 0, 0, 0, 0 } });
            zero_at += 64;
            if (zero_at >= (int)(32768 * sizeof(HeadT))) {
                chain[0] = 0xFFFF;
                phase = 3; st = kHsCtrl;                             // (blocks too short for any match go straight to the last literals)
            }
        } else if (inI) {                                            // lz4hc.c:358-373, one position
            if (ins_go) {
                const int p = next, prev = (int)v_head;
                const uint32_t delta = (p < prev || p - prev > kMaxDistance) ? (uint32_t)kMaxDistance : (uint32_t)(p - prev);
                chain_put(p, delta);
                head[hsh] = (HeadT)p;
                next = p + 1;
                if (next >= s_ip) { chain_flush(); st = kHsHead; }   // (the walk that follows may read what this burst wrote)
            }
            if (iw_load) {
                if (iw_full) { iw = v_y; iw_len = 16; } else { iw = Vec16{ { v_w, 0, 0, 0 } }; iw_len = 4; }
                iw_pos = iw_want;
            }
        } else if (inH) {
            if (!head_go) { s_word = v_w; s_wok = 1; }
            else {
                s_fwd_ok = s_ip + 4 + 16 <= matchlimit;
                s_fwd = v_y;
                s_ref = (int)v_head;
                if (phase == 0 && s_ref >= s_ip - 4) st = kHsRep;
                else st = (s_ref >= s_ip - kMaxDistance && s_ref >= 0) ? (int)kHsHop : (int)kHsCtrl;
            }
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
        } else if (inL) {                                            // lz4hc.c:437-455: c_s walks from ip to end (c_r)
            if (s_repl > 0) { c_s = s_ip; c_r = s_ip + s_repl - 3; s_repl = -1; }   // first visit: set up the walk (its loads come next iteration)
            else {
                const int q = c_s, end = c_r;
                chain_put(q, (uint32_t)s_delta);
                if (!(q < end - s_delta)) head[hash15(v_w)] = (HeadT)q;   // do { chain; head } while (q < end): runs at least once
                c_s = q + 1;
                if (!(q < end - s_delta) && c_s >= end) { chain_flush(); next = end; s_repl = 0; st = kHsCtrl; }
            }
        }
    }
}

}  // namespace lz4hip
