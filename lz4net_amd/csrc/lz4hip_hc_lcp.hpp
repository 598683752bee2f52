// lz4hip_hc_lcp.hpp -- batched LZ4HC for blocks <= 64 KiB, one LANE per block, over precomputed chains that also carry the
// number of bytes each position shares with its chain predecessor: a chain walk of the best-match search reads ONE table entry
// per candidate and the input not at all.  Bit-exact to LZ4_compressHCCtx and its match finder (original/lz4hc.c:330-755).
//
// lz4hip_hc_nat.hpp removed the insert loop; what was left was the walk -- 110 k hops per fuzzer block, each one a chain entry
// and a candidate probe in two different lines of memory, at the fabric's random-sector rate.  The reference's candidate test
// (lz4hc.c:426-431: *(ref + ml) == *(ip + ml), then A32(ref) == A32(ip), then the exact common length mlt, keep if mlt > ml) is
// a pure function of F(c) = the exact common length of the search position ip and the candidate c: the candidate is kept iff
// F(c) >= 4 and F(c) > ml.  And F along a chain follows from the table: with c' the chain predecessor of c and
// lcp[c] = the common length of c and c',
//      F(c') = min(F(c), lcp[c])   if F(c) != lcp[c],          F(c') >= F(c)   if they are equal (then compare on from there).
// The table entry of position p is  chain[p] | lcp[p] << 16 | y[p] << 24  (u32; lcp capped at 255 = "255 or more: compare on", and at
// matchlimit - p, which no search from a later position can reach; y[p] = the predecessor's byte BEHIND the shared bytes,
// in[p - chain[p] + lcp[p]] -- round 5).  The equal case F(c) == lcp[c] used to cost a 16-byte compare of the input (a third of
// the hops of fuzzer-style data, each a random sector of the block): F(c') > lcp[c] iff in[ip + lcp[c]] == y[c], and that one
// byte of the search position's side is in the lane's register window nearly always -- so the walk only compares when the
// byte does match.  Built up front for every position by two kernels:
// hc_nat_chain_kernel<uint32_t> (natural chains, lz4hip_hc_nat.hpp) and hc_lcp_fill_kernel (the block staged in LDS, one
// position per thread).  The first candidate's length is lcp[ip] itself (the entry of the search position, read before ip is
// inserted: its predecessor is the bucket's head), so the repeat detection (lz4hc.c:411-421) needs no input either.
// The repeat fill (lz4hc.c:437-455) writes  delta | min(255, ip + repl - q) << 16 | in[ip + repl - delta] << 24  for q in [ip, end):
// the run's positions share exactly the rest of the matched region with their predecessor at distance delta, and the byte
// behind it on the predecessor's side is the same for all of them.
// The wider-match search (lz4hc.c:462-518) walks the same way; its filter byte *(startLimit + longest) vs
// *(ref - delta + longest) lies inside the known common region for almost every candidate (no load), the backward extension
// reads the input as before.
#pragma once
#include "lz4hip_hc_nat.hpp"

#ifndef LZ4HIP_STAT
#define LZ4HIP_STAT(slot, cond) ((void)0)   /* the emulator build counts lane-steps per state (tools/emu_hc_stats.py) */
#endif
#ifndef LZ4HIP_STAT_ADD
#define LZ4HIP_STAT_ADD(slot, n) ((void)0)
#endif

namespace lz4hip {

constexpr size_t kHcLcpTableBytes = 65536 * sizeof(uint32_t);   // per block
constexpr int kHcLcpFillThreads = 256;
constexpr int kHcLcpFillLdsBytes = 65536 + 32;                  // the block (+ the 16-byte compares' overshoot)
constexpr int kHcLcpCap = 255;
constexpr int kHcLcpCtrlEvery = 8, kHcLcpCtrlLanes = 32;        // control-flow batching (profiles/r03/hc_chains_with_shared_lengths.txt)

// lcp[p] for every position p of block first + blockIdx.x whose chain entry hc_nat_chain_kernel<uint32_t> has written.
__global__ void __launch_bounds__(kHcLcpFillThreads) hc_lcp_fill_kernel(Batch b, long long first, uint8_t* tables)
{
    LZ4HIP_DYN_LDS(lds);
    const int tid = (int)threadIdx.x;
    const int64_t blk = (int64_t)first + blockIdx.x;
    const int n = batch_src_len(b, blk);
    if (n > 65536) return;
    const uint8_t* const in = batch_src(b, blk);
    uint32_t* const table = (uint32_t*)(tables + (size_t)blockIdx.x * kHcLcpTableBytes);
    // the block into LDS: the sixteen 16-byte loads of a thread are in flight together (addresses clamped to the block)
    {
        Vec16 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { const int i = (tid + k * kHcLcpFillThreads) * 16; v[k] = n >= 16 ? load_v16(in + (i + 16 <= n ? i : n - 16)) : Vec16{ { 0, 0, 0, 0 } }; }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = (tid + k * kHcLcpFillThreads) * 16;
            if (i + 16 <= n) wv::store16(lds + i, v[k].w[0], v[k].w[1], v[k].w[2], v[k].w[3]);
            else for (int j = i; j < n; j++) lds[j] = in[j];
        }
    }
    wv::block_sync();
    const uint32_t* const lw = (const uint32_t*)lds;
    const int matchlimit = n - kLastLiterals;
    // Eight rows of 256 consecutive positions per round; the entries of the next round are loaded (addresses clamped, no
    // branches) while this round's lengths are counted in LDS.
    //  * Inside a match consecutive positions have consecutive predecessors, and then their lengths are consecutive too: if
    //    prev(p) == prev(p - 1) + 1, position p shares exactly one byte less with its predecessor than p - 1 does (as long as
    //    that is at least one byte).  So only the first lane of such a run (a LEADER) counts bytes -- up to 255 + 63, so that a
    //    whole wavefront of followers can be served from one count -- and the others take its result through a shuffle; a
    //    follower past the end of its leader's common bytes counts for itself afterwards.
    //  * The counts of a thread's rows run as ONE loop over its queue of rows (16 bytes per step): a wavefront then iterates
    //    max-over-lanes of the SUM of their counts' steps, not the sum over the rows of the longest count among 64 lanes.
    constexpr int U = 8;
    const int lastp = n - 4;
    if (lastp < 1) return;
    const int lane = tid & 63;
    const uint64_t lanes_le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    auto entry_of = [&](int p) -> uint32_t { return table[p <= lastp ? p : lastp]; };
    uint32_t e[U], e2[U], r[U];
    auto pick = [&](const uint32_t (&v)[U], int u) -> uint32_t {
        uint32_t x = v[U - 1];
#pragma unroll
        for (int k = U - 2; k >= 0; k--) x = u == k ? v[k] : x;
        return x;
    };
    // common bytes of in[p..] and in[p - chain..] for the rows of `todo` (bit u = row u of this round), at most room / 255 + extra
    auto count_rows = [&](uint32_t todo, int p0, int extra) {
        int u = 0, p = 0, prev = 0, cap = 0, off = 0;
        bool fresh = true;
        while (todo) {
            if (fresh) {
                u = __builtin_ctz(todo);
                p = p0 + u * kHcLcpFillThreads;
                prev = p - (int)(pick(e, u) & 0xFFFFu);
                cap = matchlimit - p;
                cap = cap < 0 ? 0 : (cap > kHcLcpCap + extra ? kHcLcpCap + extra : cap);
                off = 0; fresh = false;
            }
            bool done = off >= cap;
            if (!done) {                                             // 16 bytes of both sides: five aligned dwords each
                const int xa = p + off, xb = prev + off;
                const uint32_t* const wa = lw + (xa >> 2);
                const uint32_t* const wb = lw + (xb >> 2);
                const uint32_t sa = (uint32_t)xa & 3u, sb = (uint32_t)xb & 3u;
                const uint32_t a0 = wa[0], a1 = wa[1], a2 = wa[2], a3 = wa[3], a4 = wa[4];
                const uint32_t b0 = wb[0], b1 = wb[1], b2 = wb[2], b3 = wb[3], b4 = wb[4];
                const uint32_t d0 = wv::alignbyte(a1, a0, sa) ^ wv::alignbyte(b1, b0, sb), d1 = wv::alignbyte(a2, a1, sa) ^ wv::alignbyte(b2, b1, sb);
                const uint32_t d2 = wv::alignbyte(a3, a2, sa) ^ wv::alignbyte(b3, b2, sb), d3 = wv::alignbyte(a4, a3, sa) ^ wv::alignbyte(b4, b3, sb);
                if (d0 | d1) { off += d0 ? (__builtin_ctz(d0) >> 3) : 4 + (__builtin_ctz(d1) >> 3); done = true; }
                else if (d2 | d3) { off += d2 ? 8 + (__builtin_ctz(d2) >> 3) : 12 + (__builtin_ctz(d3) >> 3); done = true; }
                else { off += 16; done = off >= cap; }
            }
            if (done) {
                off = off > cap ? cap : off;
#pragma unroll
                for (int k = 0; k < U; k++) r[k] = u == k ? (uint32_t)off : r[k];
                todo &= todo - 1;
                fresh = true;
            }
        }
    };
#pragma unroll
    for (int u = 0; u < U; u++) e[u] = entry_of(1 + tid + u * kHcLcpFillThreads);
    for (int p0 = 1 + tid; p0 - tid <= lastp; p0 += U * kHcLcpFillThreads) {       // (whole wavefronts: the shuffles need every lane)
        uint32_t leader_rows = 0, follower_rows = 0, late_rows = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            e2[u] = entry_of(p0 + (U + u) * kHcLcpFillThreads);
            r[u] = 0;
            const int p = p0 + u * kHcLcpFillThreads;
            const int prev = p - (int)(e[u] & 0xFFFFu);
            const int prev_left = (int)wv::shuffle((uint32_t)prev, lane - 1);
            const bool valid = p <= lastp;
            const bool follower = valid & (lane > 0) & (prev == prev_left + 1);
            leader_rows |= (valid & !follower) ? 1u << u : 0u;
            follower_rows |= follower ? 1u << u : 0u;
        }
        count_rows(leader_rows, p0, 63);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t leaders = wv::ballot(((follower_rows >> u) & 1u) == 0u);
            const int a = 63 - __builtin_clzll(leaders & lanes_le);   // my leader (myself if I am one; lane 0 always is)
            const int from_leader = (int)wv::shuffle(r[u], a) - (lane - a);
            if ((follower_rows >> u) & 1u) {
                if (from_leader >= 1) r[u] = (uint32_t)from_leader;
                else late_rows |= 1u << u;
            }
        }
        count_rows(late_rows, p0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = p0 + u * kHcLcpFillThreads;
            int room = matchlimit - p;
            room = room < 0 ? 0 : (room > kHcLcpCap ? kHcLcpCap : room);
            const uint32_t len = r[u] > (uint32_t)room ? (uint32_t)room : r[u];
            // the predecessor's byte behind the shared bytes (only looked at when len is a real mismatch distance: a length at
            // one of its caps never takes part in the walk's equal case)
            const int yp = p - (int)(e[u] & 0xFFFFu) + (int)len;
            const uint32_t y = lds[yp < 0 ? 0 : (yp < n ? yp : n - 1)];
            if (p <= lastp) table[p] = (e[u] & 0xFFFFu) | (len << 16) | (y << 24);
        }
#pragma unroll
        for (int k = 0; k < U; k++) e[k] = e2[k];
    }
}

enum HcLcpState {
    kLsFetch = 0,   // take the next block from the counter
    kLsHead,        // the entry of the search position: its bucket's head and the length shared with it
    kLsHop,         // one candidate: evaluate it, read its entry, derive the next candidate's length
    kLsCmp,         // exact length where the table only gives a lower bound, 16 bytes per step
    kLsBack,        // wider match only: backward extension, up to 4 bytes per step
    kLsRepl,        // best match only: repeat optimisation fill, up to four entries per step
    kLsCtrl,        // search complete: control flow up to the next search (or the end of the block)
    kLsExit
};

// ctrl_every / ctrl_lanes: the control flow runs for all waiting lanes every ctrl_every-th iteration (a power of two) or as soon as
// ctrl_lanes lanes wait (lz4hip_hc_parse.inc).
__global__ void __launch_bounds__(64) encode_hc_lcp_kernel(Batch b, long long first, long long count, unsigned long long* counter, uint8_t* tables,
                                                           int ctrl_every, int ctrl_lanes)
{
    // the names the shared control flow (lz4hip_hc_parse.inc) uses for its states and its batching
    constexpr int kHsCtrl = kLsCtrl, kHsFetch = kLsFetch;
    const int kHcCtrlEvery = ctrl_every, kHcCtrlBatchLanes = ctrl_lanes;
    // ---- the block ----
    const uint8_t* in = nullptr;
    uint8_t* out = nullptr;
    uint32_t* table = nullptr;
    int64_t blk = 0;
    int n = 0, cap = 0, mflimit = 0, matchlimit = 0;
    // ---- the parse (variables of LZ4_compressHCCtx, lz4hc.c:553-742) ----
    int ip = 0, anchor = 0, op = 0;
    int ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0 = 0, ref0 = 0;
    int ml = 0, ml2 = 0, ml3 = 0, ml0 = 0;
    int phase = 0;                 // which search the control flow is waiting for: 0 best (ip), 1 wider -> ml2, 2 wider -> ml3
    // ---- the search in progress ----
    int st = kLsFetch;
    int s_ip = 0, s_limit = 0, s_back = 0;      // position searched, start limit (wider), ip - start_limit
    int s_len = 0;                 // best length so far (ml / longest)
    int s_match = 0, s_start = 0;  // where it was found (and, wider, where it starts)
    int s_ref = 0;                 // the candidate
    int s_f = 0;                   // F(s_ref): exact common length of in[s_ip..] and in[s_ref..], capped at matchlimit - s_ip
    int s_first = 0;               // s_ref is the bucket's head (the repeat detection looks at it, lz4hc.c:411)
    int s_link = 0, s_lcp = 0;     // entry of s_ref (kept across the backward extension)
    uint32_t s_y = 0, s_ry = 0;    // ... its predecessor's byte behind the shared bytes; the same byte for the entries of a repeat fill
    int attempts = 0;
    int stat_hops = 0;             // (emulator statistics only: candidates evaluated by the search in progress; dead code in product builds)
    uint32_t s_probe = 0;
    int s_probe_ok = 0;            // s_probe is in[start_limit + longest] for the current s_len (wider)
    int s_repl = 0, s_delta = 0;
    int c_n = 0, c_s = 0, c_r = 0, c_fwd_end = 0;
    // The search position's side of the length counts comes from a 32-byte register window of the input (searches follow each
    // other a few bytes apart and count on from 4 .. 20 bytes): bytes [w_pos, w_pos + 32) of the block, refilled by the count
    // that misses it.
    uint32_t iw0 = 0, iw1 = 0, iw2 = 0, iw3 = 0, iw4 = 0, iw5 = 0, iw6 = 0, iw7 = 0;
    int w_pos = -64;
    auto win16 = [&](int o) -> Vec16 {                               // bytes [o, o + 16) of the window, 0 <= o <= 16
        const int q = o >> 2;
        const uint32_t sh = (uint32_t)o & 3u;
        auto pick = [&](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4) { return q < 2 ? (q == 0 ? x0 : x1) : (q == 2 ? x2 : (q == 3 ? x3 : x4)); };
        const uint32_t a0 = pick(iw0, iw1, iw2, iw3, iw4), a1 = pick(iw1, iw2, iw3, iw4, iw5), a2 = pick(iw2, iw3, iw4, iw5, iw6),
                       a3 = pick(iw3, iw4, iw5, iw6, iw7), a4 = pick(iw4, iw5, iw6, iw7, 0u);
        return Vec16{ { wv::alignbyte(a1, a0, sh), wv::alignbyte(a2, a1, sh), wv::alignbyte(a3, a2, sh), wv::alignbyte(a4, a3, sh) } };
    };

    auto win_byte = [&](int o) -> uint32_t {                         // byte o of the window, 0 <= o < 32
        // (selects that STAY selects, wv::sel: written as nested ?: over eight scalars the compiler builds an indexed array in scratch memory)
        const wv::mask_t m1 = wv::cond((o & 4) != 0), m2 = wv::cond((o & 8) != 0), m4 = wv::cond((o & 16) != 0);
        const uint32_t lo = wv::sel(m2, wv::sel(m1, iw3, iw2), wv::sel(m1, iw1, iw0)), hi = wv::sel(m2, wv::sel(m1, iw7, iw6), wv::sel(m1, iw5, iw4));
        return (wv::sel(m4, hi, lo) >> (8u * ((uint32_t)o & 3u))) & 255u;
    };

    auto request = [&](int pos, int start_limit, int longest, int match0, int start0_) {
        s_ip = pos; s_limit = start_limit; s_back = pos - start_limit; s_len = longest; s_match = match0; s_start = start0_;
        attempts = kHcAttempts; s_repl = 0; s_delta = 0; s_probe_ok = 0;
        stat_hops = 0;
        st = kLsHead;
    };

    int it = 0;
    for (;;) {
        if (st == kLsFetch) {
            const long long k = (long long)atomicAdd(counter, 1ull);
            if (k >= count) st = kLsExit;
            else {
                blk = (int64_t)first + k;
                table = (uint32_t*)(tables + (size_t)k * kHcLcpTableBytes);
                n = batch_src_len(b, blk); cap = batch_dst_cap(b, blk);
                in = batch_src(b, blk); out = batch_dst(b, blk);
                if (n > 65536) { b.result[blk] = -2000000002; st = kLsFetch; }   // LZ4HIP_E_ARGUMENT: this launch is for blocks <= 64 KiB
                else {
                    mflimit = n - kMfLimit; matchlimit = n - kLastLiterals;
                    ip = 1; anchor = 0; op = 0;                                  // lz4hc.c:581
                    phase = 3; st = kLsCtrl; w_pos = -64;
                }
            }
        }
        if (!wv::any(st != kLsExit)) break;

#include "lz4hip_hc_parse.inc"

        // ================= one memory step of the state each lane is in =================
        const bool inH = st == kLsHead, inP = st == kLsHop, inC = st == kLsCmp, inB = st == kLsBack, inL = st == kLsRepl;
        LZ4HIP_STAT(0, st != kLsExit); LZ4HIP_STAT(1, inH); LZ4HIP_STAT(2, inP); LZ4HIP_STAT(3, inC); LZ4HIP_STAT(4, inB); LZ4HIP_STAT(5, inL); LZ4HIP_STAT(6, st == kLsCtrl);
        const int f_a = s_ip + c_n, f_b = s_ref + c_n;
        const bool cmp16 = inC & (f_a + 16 <= matchlimit);
        const bool back4 = inB & (c_s - s_limit >= 4) & (c_r >= 4);
        // wider match: the filter byte *(startLimit + longest) vs *(ref - delta + longest) (lz4hc.c:478) is index j of the
        // pair (ip, ref): inside the common region it is equal, at its end (a real mismatch) different, elsewhere it is read
        const int fj = s_len - s_back;
        const bool wide = inP & (phase != 0) & (s_f >= kMinMatch);
        const bool f_pass = (fj >= 0) & (fj < s_f), f_fail = (fj == s_f) & (s_f < matchlimit - s_ip);
        const bool f_read = wide & !f_pass & !f_fail;
        // (1) table entry: of the search position / of the candidate
        uint32_t v_e = 0;
        if (inH | inP | inC) v_e = table[inH ? s_ip : s_ref];
        // (2, 3) wider match: the candidate's filter byte; the search position's when the best length has changed
        uint32_t v_cb = 0, v_pb = 0;
        if (f_read) v_cb = in[s_ref - s_back + s_len];
        if (f_read & (s_probe_ok == 0)) v_pb = in[s_limit + s_len];
        // (4, 5) backward extension: 4 bytes before both starts
        uint32_t v_w = 0, v_w2 = 0;
        if (back4) { v_w = load_u32(in + c_s - 4); v_w2 = load_u32(in + c_r - 4); }
        // (6, 7) exact length: 16 bytes of both sides
        Vec16 v_x = { { 0, 0, 0, 0 } }, v_y = { { 0, 0, 0, 0 } }, v_x2 = { { 0, 0, 0, 0 } };
        const bool w_hit = (f_a >= w_pos) & (f_a + 16 <= w_pos + 32);
        const bool w_fill = cmp16 & !w_hit & (f_a + 32 <= n);        // (the miss also fetches the following 16 bytes: the new window)
        if (cmp16) v_y = load_v16(in + f_b);
        if (cmp16 & !w_hit) v_x = load_v16(in + f_a);
        if (w_fill) v_x2 = load_v16(in + f_a + 16);
        // (8) repeat fill, first step: the byte behind the run on its predecessor's side (what its entries carry as y)
        uint32_t v_rb = 0;
        if (inL & (s_repl > 0)) v_rb = in[s_ip + s_repl - s_delta];

        // ---- process ----
        bool adv = false;                                            // the walk moves on to the next candidate (below, once)
        if (inH) {                                                   // HASH_POINTER(ip) after LZ4HC_Insert(ip) == ip - natural chain[ip]
            s_ref = s_ip - (int)(v_e & 0xFFFFu);
            s_f = (int)((v_e >> 16) & 255u);
            s_first = 1;
            if (s_f >= kHcLcpCap) { c_n = kHcLcpCap; st = kLsCmp; }
            else st = kLsHop;                                        // (blocks <= 64 KiB: the head is always within MAX_DISTANCE)
        }
        // kLsCmp and kLsHop: a length count that completes in this step goes straight on to the candidate's evaluation -- the
        // candidate's entry was loaded with the bytes that were compared (a third of the hops used to take two iterations)
        bool hop_now = inP;
        if (inC) {                                                   // common length of in[s_ip + c_n ..] and in[s_ref + c_n ..] up to matchlimit
            int add = 0;
            bool more = false;
            if (cmp16) {
                if (w_hit) v_x = win16(f_a - w_pos);
                else if (w_fill) {
                    iw0 = v_x.w[0]; iw1 = v_x.w[1]; iw2 = v_x.w[2]; iw3 = v_x.w[3]; iw4 = v_x2.w[0]; iw5 = v_x2.w[1]; iw6 = v_x2.w[2]; iw7 = v_x2.w[3];
                    w_pos = f_a;
                }
                const uint64_t d0 = (v_x.w[0] ^ v_y.w[0]) | ((uint64_t)(v_x.w[1] ^ v_y.w[1]) << 32);
                const uint64_t d1 = (v_x.w[2] ^ v_y.w[2]) | ((uint64_t)(v_x.w[3] ^ v_y.w[3]) << 32);
                if (d0) add = __builtin_ctzll(d0) >> 3;
                else if (d1) add = 8 + (__builtin_ctzll(d1) >> 3);
                else { add = 16; more = true; }
            } else {                                                 // the last bytes before matchlimit, one by one (rare)
                while (f_a + add < matchlimit && in[f_a + add] == in[f_b + add]) add++;
            }
            c_n += add;
            if (!more) {
                s_f = c_n;
                // (wider match: unless the filter byte has to be read -- then the candidate takes its own kLsHop step)
                const bool known = (s_f < kMinMatch) | ((fj >= 0) & (fj < s_f)) | ((fj == s_f) & (s_f < matchlimit - s_ip));
                if ((phase == 0) | known) hop_now = true;
                else st = kLsHop;
            }
        }
        if (hop_now) {
            stat_hops++;
            s_link = (int)(v_e & 0xFFFFu); s_lcp = (int)((v_e >> 16) & 255u); s_y = v_e >> 24;
            if (phase == 0) {
                if (s_first && s_ref >= s_ip - 4) {                  // lz4hc.c:411-421: not one of the attempts
                    if (s_f >= kMinMatch) { s_delta = (s_ip - s_ref) & 0xFFFF; s_repl = s_len = s_f; s_match = s_ref; }
                } else {                                             // lz4hc.c:424-434
                    attempts--;
                    if (s_f >= kMinMatch && s_f > s_len) { s_len = s_f; s_match = s_ref; }
                }
                adv = true;
            } else {                                                 // lz4hc.c:474-516
                attempts--;
                bool pass = false;
                if (s_f >= kMinMatch) {
                    if (f_read) {                                    // (kLsHop lanes only: the bytes were loaded above)
                        if (!s_probe_ok) { s_probe = v_pb; s_probe_ok = 1; }
                        pass = v_cb == s_probe;
                    } else pass = (fj >= 0) & (fj < s_f);
                }
                if (pass) { c_fwd_end = s_ip + s_f; c_s = s_ip; c_r = s_ref; st = kLsBack; }
                else adv = true;
            }
        }
        if (inB) {
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
                adv = true;
            }
        } else if (inL) {                                            // lz4hc.c:437-455: DELTANEXT(q) = delta for q in [ip, end)
            if (s_repl > 0) { c_s = s_ip; c_r = s_ip + s_repl - 3; c_fwd_end = s_ip + s_repl; s_repl = -1; s_ry = v_rb; }
            int q = c_s;
            const uint32_t d = (uint32_t)s_delta | (s_ry << 24);
            auto entry = [&](int at) -> uint32_t { const int l = c_fwd_end - at; return d | ((uint32_t)(l > kHcLcpCap ? kHcLcpCap : l) << 16); };
            if ((q & 3) == 0 && q + 4 <= c_r) {
                store_v16((uint8_t*)(table + q), Vec16{ { entry(q), entry(q + 1), entry(q + 2), entry(q + 3) } });
                q += 4;
            } else {
                table[q] = entry(q);
                q++;
            }
            c_s = q;
            if (q >= c_r) { s_repl = 0; st = kLsCtrl; }
        }
        // the next candidate of the walk: c' = c - chain[c], F(c') from F(c) and lcp[c]; ends the search when the walk is over
        if (adv) {
            const int c2 = s_ref - s_link;
            if (!(c2 >= s_ip - kMaxDistance && attempts > 0 && c2 >= 0)) {
                st = (s_repl && phase == 0) ? (int)kLsRepl : (int)kLsCtrl;
                // (what a bucket-contiguous table would need for this search: its own entry + the candidates' entries, four per 16-byte read)
                LZ4HIP_STAT_ADD(12, 1); LZ4HIP_STAT_ADD(13, stat_hops); LZ4HIP_STAT_ADD(14, (stat_hops + 1 + 3) / 4); LZ4HIP_STAT_ADD(15, (stat_hops + 1 + 7) / 8);
                LZ4HIP_STAT(16, stat_hops <= 1); LZ4HIP_STAT(17, stat_hops >= 2 && stat_hops <= 3); LZ4HIP_STAT(18, stat_hops >= 4 && stat_hops <= 7); LZ4HIP_STAT(19, stat_hops >= 8 && stat_hops <= 31); LZ4HIP_STAT(20, stat_hops >= 32);
            }
            else {
                const int l = s_lcp;
                s_ref = c2; s_first = 0;
                LZ4HIP_STAT(7, true);
                if (s_f < l) st = kLsHop;                                        // F(c') = F(c)
                else if (s_f > l && l < kHcLcpCap) { s_f = l; st = kLsHop; }     // F(c') = lcp[c]
                else {
                    // equal, or both >= 255: F(c') is at least l.  Equal and below the cap: it is MORE than l only if the search
                    // position's byte l equals the predecessor's (y of the entry just left) -- a look at the register window
                    // instead of a 16-byte compare; at matchlimit - s_ip nothing can be added either.
                    const int pos = s_ip + l;
                    bool exact = false;
                    if (l < kHcLcpCap) {
                        if (pos >= matchlimit) exact = true;
                        else if ((pos >= w_pos) & (pos < w_pos + 32)) exact = win_byte(pos - w_pos) != s_y;
                    }
                    LZ4HIP_STAT(8, true); LZ4HIP_STAT(9, l < kHcLcpCap); LZ4HIP_STAT(10, (l < kHcLcpCap) & (pos >= w_pos) & (pos < w_pos + 32)); LZ4HIP_STAT(11, exact);
                    if (exact) st = kLsHop;                                      // F(c') = l = F(c)
                    else { c_n = l; st = kLsCmp; }                               // compare on
                }
            }
        }
        // A best-match search that is complete needs no sequence encode to know the next search (lz4hc.c:586-597: no match ->
        // the next position; a match -> the wider search at ip + ml - 2): start it now instead of waiting for the batched control
        // flow (three searches in four are best-match searches; waiting lanes were a third of all lane-iterations).
        if ((st == kLsCtrl) & (phase == 0)) {
            ml = s_len; ref = s_match;
            if (!ml) {
                ip++;
                if (ip < mflimit) request(ip, ip, 0, ref, 0);
                else phase = 3;                                      // (the control flow finishes the block)
            } else {
                start0 = ip; ref0 = ref; ml0 = ml;
                if (ip + ml < mflimit) { phase = 1; request(ip + ml - 2, ip + 1, ml, ref2, start2); }
                else phase = 4;
            }
        }
    }
}

}  // namespace lz4hip
