// (NOT part of the product library: the third generation of the lane decoder, superseded by lz4net_amd/csrc/lz4hip_decode_lane4.hpp;
//  kept under tools/ab/ for A/B runs -- libraries built with -DLZ4HIP_TUNING_BUILD -- and emulator tests)
// lz4hip_decode_lane3.hpp -- lane-per-block LZ4 decoder, third generation: the same mapping and the same LDS rings as
// lz4hip_decode_lane.hpp (64 blocks per wavefront, input staged through a per-lane ring, output appended to a per-lane
// ring and flushed in 64-byte lines by four lanes per line), rebuilt around what the round-2 counters said
// (profiles/r02/pmc_decode_*.json):
//
//   * WAITING.  The second generation issued its far-match fetch late in an iteration and consumed it early in the next
//     one -- two thirds of an iteration of cover for a fabric round trip that takes longer than a whole iteration under
//     load (38 % of the wave-cycles parked at s_waitcnt on fuzzer data, 64 % on record data).  Here an iteration has a TOP
//     (cooperative flush, parse, far fetch, staging request) and a BOTTOM (wait, land staged input, append, promote); a
//     fetch leaves at the top of iteration i and is consumed at the bottom of iteration i+1: more than one and a half
//     iterations.  That only works if the wait at the bottom of i+1 does NOT also wait for the loads issued at the top of
//     i+1, so every vector-memory instruction of the loop is issued through wv::vm_load16_pred / vm_store16_pred (inline
//     assembly, predicated inside, always issued): exactly FOUR per iteration (two flush stores, the far fetch, the
//     staging load), and the one wait is s_waitcnt vmcnt(4) (wv::vm_wait<4>).  Rare paths that issue further,
//     compiler-visible accesses can only make that wait stricter (everything returns in issue order).
//   * INSTRUCTIONS.  Appends no longer carry the dword that contains the output cursor in a register and read it back
//     after every append: the first dword of an append is merged into the ring with DS_MSKOR_B32 (wv::lds_mskor).  The
//     offset field of a sequence is read from the staging ring at its byte position instead of being picked out of four
//     registers; the last bytes of the source are served by the staging ring like all others (no byte-wise literal
//     path), and everything a well-formed stream needs once per block or less -- length bytes of 255, the final literal
//     run, every error -- is decided by a byte-wise parser behind ONE wave-level branch (`trap`).
//   * RING SIZE.  The ring may have any multiple of 16 bytes (row wrap by unsigned min instead of a mask), so that LDS per
//     wavefront -- which sets the residency -- can be traded against the share of matches that have to be fetched from
//     global memory in steps finer than a factor of two: the kernel is bound by those fetches (each drags a line across
//     the fabric for 16 useful bytes), not by arithmetic.
//
// Same functions / return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).
#pragma once
#include "lz4hip_common.hpp"

#ifndef LZ4HIP_ITERATION_HOOK
#define LZ4HIP_ITERATION_HOOK(lane) ((void)0)
#endif

namespace lz4hip {

#ifndef LZ4HIP_DEC3_FLUSH_RECS
#define LZ4HIP_DEC3_FLUSH_RECS 32     /* the emulator's 'starved' build: 4 */
#endif
constexpr int kL3FlushRecs = LZ4HIP_DEC3_FLUSH_RECS;      // lines stored per iteration (two store instructions of 16 lines)
constexpr int kL3LoadRecs = 32;                           // input pieces requested per iteration (one load instruction)
constexpr unsigned lane3_lds_bytes(int ring_bytes, int stage_bytes)
{
    return 64u * (unsigned)(ring_bytes + stage_bytes) + 16u * (unsigned)(32 + kL3LoadRecs);
}

enum L3Kind { kK3None = 0, kK3Near = 1, kK3Far = 2, kK3Lit = 3, kK3Zero = 4 };
enum L3Flag { kF3Final = 1, kF3Err = 2, kF3Header = 4 };   // pending sequence: final literal run / corrupt stream / no match yet (its header follows the literals)

// All 64 lanes of the wavefront call this together and stay in the loop until the last one is done.
// POL: cache policy of the loads (wv::vm_load16_pred): low two bits = far-match fetches, next two bits = input pieces
template <bool KNOWN, int R, int SB, int POL = 0>
LZ4HIP_DEVICE int lane3_decode_block(unsigned char* lds, int lane, bool active, const uint8_t* __restrict__ src, int iend,
                                     uint8_t* dst, int oend)
{
    static_assert(R >= 128 && R % 16 == 0 && R <= 1008, "ring: a multiple of 16 bytes, 128 .. 1008");
    static_assert(SB == 64 || SB == 128, "staging ring: 64 or 128 bytes per lane");
    constexpr int RW = R / 4;                                        // ring rows (one dword per lane per row)
    constexpr bool RPOW2 = (RW & (RW - 1)) == 0;
    constexpr uint32_t kRingBytes = (uint32_t)RW * 256u;             // the 64 rings, dword-interleaved: row r of lane l at r * 256 + l * 4
    constexpr int PIECE = SB / 2, HELPERS = PIECE / 16;
#ifdef LZ4HIP_DEC3_LOAD_PIECES
    constexpr int PIECES_PER_LOAD = LZ4HIP_DEC3_LOAD_PIECES;         // the emulator's 'starved' build: 2
#else
    constexpr int PIECES_PER_LOAD = 64 / HELPERS;
#endif
    constexpr uint32_t kStageBytes = 64u * SB, kStageMask = kStageBytes - 1u;
    constexpr int kNearMax = R - 20;                                 // an append writes whole dwords, up to 19 bytes past its last byte
    constexpr int kFlushUrgent = R >= 256 ? 128 : 64;
    unsigned char* const stage = lds + kRingBytes;
    Aligned16* const flush_rec = (Aligned16*)(stage + kStageBytes);
    Aligned16* const load_rec = flush_rec + 32;
    const uint32_t lane4 = (uint32_t)lane << 2;

    // ring-relative byte address (row * 256 | lane * 4) plus k rows, wrapped
    auto ring_add = [](uint32_t a, uint32_t rows256) -> uint32_t {
        const uint32_t t = a + rows256;
        if (RPOW2) return t & (kRingBytes - 1u);
        const uint32_t u = t - kRingBytes;                           // (underflows unless t ran past the last row)
        return u < t ? u : t;
    };
#define L3_RING(a) (*(uint32_t*)(lds + (a)))
#define L3_STAGE(a) (*(const uint32_t*)(stage + (a)))
#define L3_PHASE_SEL(p_) wv::alignbyte(0x07060504u, 0x03020100u, (uint32_t)(p_) & 3u)

    // ---- per-lane state ----
    const int skew = (int)((uint64_t)src & (uint64_t)(PIECE - 1));
    const uint64_t src_al = (uint64_t)src - (uint64_t)skew;
    const int in_total = iend > 0 ? (int)(((int64_t)skew + iend + PIECE - 1) & ~(int64_t)(PIECE - 1)) : 0;
    int ip = 0;                  // input cursor (block coordinates): next token / next streamed literal / next header
    int have = 0;                // aligned coordinates (ip + skew): bytes [have - SB, have) are staged
    int pend_a = 0, pend_b = 0;  // a piece of this lane is in flight (by the parity of the iteration that requested it)
    int op = 0, fl = 0;          // bytes produced / bytes stored to dst (multiple of 64)
    uint32_t oa = lane4;         // ring address of the dword that contains op
    int kind = kK3None, rem = 0, off = 8;
    int gready = 0;              // kK3Far: the 16 bytes fetched in the previous iteration are this lane's next chunk
    // parsed-ahead sequence
    int pv = 0, p_ll = 0, p_st = 0, p_ml = 0, p_off = 0, p_flags = 0, p_res = 0;
    uint32_t p_l0 = 0, p_l1 = 0, p_l2 = 0;
    int hdr = 0;                 // the cursor is at a sequence's offset field (its literals were streamed)
    uint32_t token = 0;
    int final_seen = 0, final_run = 0, result = 0, done = 0, flush_blocked = 0;
    if (!active || (!KNOWN && iend == 0)) { done = 1; final_seen = 1; }   // lz4.c:946 returns -(0)
    // cooperative piece loads (helper role), by iteration parity: data, destination in the staging area
    wv::u32x4 fa = { 0, 0, 0, 0 }, fb = { 0, 0, 0, 0 }, ha = { 0, 0, 0, 0 }, hb = { 0, 0, 0, 0 };
    uint32_t h_dst_a = 0, h_dst_b = 0;
    int h_valid_a = 0, h_valid_b = 0;
    bool inflight_a = false, inflight_b = false;                     // wave-uniform

    // Append the low n_ bytes of the data dwords at op: rotated to the byte phase of op (one v_perm_b32 per dword), the
    // first dword merged into the ring under a byte mask, the others stored whole.
#define L3_APPEND(d0_, d1_, d2_, d3_, n_, FOUR_)                                                         \
    do {                                                                                                \
        const uint32_t sb_ = (uint32_t)op & 3u;                                                         \
        const uint32_t s_ = wv::alignbyte(0x08070605u, 0x04030201u, sb_ ^ 3u);                          \
        const uint32_t a1_ = ring_add(oa, 256u), a2_ = ring_add(oa, 512u), a3_ = ring_add(oa, 768u);    \
        wv::lds_mskor(&L3_RING(oa), 0xFFFFFFFFu << (8u * sb_), wv::perm(d0_, 0u, s_));                  \
        L3_RING(a1_) = wv::perm(d1_, d0_, s_);                                                          \
        L3_RING(a2_) = wv::perm(d2_, d1_, s_);                                                          \
        if (FOUR_) {                                                                                    \
            L3_RING(a3_) = wv::perm(d3_, d2_, s_);                                                      \
            L3_RING(ring_add(oa, 1024u)) = wv::perm(0u, d3_, s_);                                       \
        } else {                                                                                        \
            L3_RING(a3_) = wv::perm(0u, d2_, s_);                                                       \
        }                                                                                               \
        oa = ring_add(oa, ((sb_ + (uint32_t)(n_)) << 6) & 0x700u);                                      \
        op += (n_);                                                                                     \
    } while (0)

    // One iteration.  ldF / ldH: the far-match and staging registers loaded in THIS iteration; usF / usH: those loaded
    // in the previous one (consumed at the bottom of this one).
    auto iteration = [&](wv::u32x4& ldF, wv::u32x4& usF, wv::u32x4& ldH, uint32_t& ld_hdst, int& ld_hvalid, bool& ld_inflight, int& ld_pend,
                         wv::u32x4& usH, uint32_t& us_hdst, int& us_hvalid, bool& us_inflight, int& us_pend) __attribute__((always_inline)) -> bool {
        LZ4HIP_ITERATION_HOOK(lane);
        // ================================ TOP ================================
        // ---- (T1) the 16 bytes at the input cursor, the 16 bytes at the source of the current near match ----
        const int A = ip + skew;
        const uint32_t xa = (((uint32_t)A << 6) & kStageMask & ~0xFFu) | lane4;
        const uint32_t z0 = L3_STAGE(xa), z1 = L3_STAGE((xa + 256u) & kStageMask), z2 = L3_STAGE((xa + 512u) & kStageMask),
                       z3 = L3_STAGE((xa + 768u) & kStageMask), z4 = L3_STAGE((xa + 1024u) & kStageMask);
        const uint32_t sx = L3_PHASE_SEL(A);
        const uint32_t x0 = wv::perm(z1, z0, sx), x1 = wv::perm(z2, z1, sx), x2 = wv::perm(z3, z2, sx), x3 = wv::perm(z4, z3, sx);
        const bool staged16 = (have - A >= 16) | (have >= in_total);  // everything the next 16 bytes can legitimately use is staged
        uint32_t v0, v1, v2, v3;
        {
            // row of output byte op - off: (op >> 2) - ((off - (op & 3) + 3) >> 2) rows back from oa
            const uint32_t offn = kind == kK3Near ? (uint32_t)off : 4u;        // (any other kind: some valid row)
            const uint32_t back = ((offn + 3u - ((uint32_t)op & 3u)) << 6) & ~0xFFu;
            uint32_t sa;
            if (RPOW2) sa = (oa - back) & (kRingBytes - 1u);
            else { const uint32_t t = oa - back, u = t + kRingBytes; sa = u < t ? u : t; }
            const uint32_t r0 = L3_RING(sa), r1 = L3_RING(ring_add(sa, 256u)), r2 = L3_RING(ring_add(sa, 512u)),
                           r3 = L3_RING(ring_add(sa, 768u)), r4 = L3_RING(ring_add(sa, 1024u));
            const uint32_t sr = L3_PHASE_SEL((uint32_t)op - offn);
            v0 = wv::perm(r1, r0, sr); v1 = wv::perm(r2, r1, sr); v2 = wv::perm(r3, r2, sr); v3 = wv::perm(r4, r3, sr);
        }

        {
        // ---- (T2) flush finished output, 64 bytes at a time, four lanes per line: ALWAYS two store instructions ----
        {
            const bool need = (done == 0) & (op - fl >= 64);
            const bool urgent = need & ((op - fl >= kFlushUrgent) | (flush_blocked != 0) | ((final_run != 0) & (rem == 0)));
            const uint64_t needy = wv::ballot(need);
            const int cnt_all = wv::popc64(needy);
            const bool go = cnt_all >= 16 || wv::any(urgent);        // wave-uniform
            int cnt = 0;
            bool mine = false;
            if (go) {
                cnt = cnt_all < kL3FlushRecs ? cnt_all : kL3FlushRecs;
                const int frank = wv::rank_below(needy);
                mine = need & (frank < kL3FlushRecs);
                if (mine) {
                    const uint64_t dp = (uint64_t)dst;
                    // ring address of the line: fl is a multiple of 64, so the line starts 16 * k rows before oa's row
                    flush_rec[frank] = Aligned16{ { ring_add(oa, kRingBytes - ((((uint32_t)(op - fl)) << 6) & ~0xFFu)), (uint32_t)fl, (uint32_t)dp, (uint32_t)(dp >> 32) } };
                }
                wv::mem_sync();
            }
            const int sub = lane & 3;
#pragma unroll
            for (int base = 0; base < 32; base += 16) {
                const int idx = base + (lane >> 2);
                const bool act = idx < cnt;
                uint64_t g = 0;
                uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                if (act) {
                    const Aligned16 r = flush_rec[idx];
                    g = ((uint64_t)r.w[2] | ((uint64_t)r.w[3] << 32)) + (uint64_t)(r.w[1] + 16u * (uint32_t)sub);
                    // the owner's lane bits are in r.w[0]; this helper takes rows 4*sub .. 4*sub+3 of the line
                    const uint32_t b0 = ring_add(r.w[0], 1024u * (uint32_t)sub);
                    q0 = L3_RING(b0); q1 = L3_RING(ring_add(b0, 256u)); q2 = L3_RING(ring_add(b0, 512u)); q3 = L3_RING(ring_add(b0, 768u));
                }
                wv::vm_store16_pred(act, g, q0, q1, q2, q3);
            }
            if (go) {
                wv::mem_sync();                                      // records and ring rows are free to be overwritten again
                fl += mine ? 64 : 0;
            }
        }
        }
        // ---- (T3) size of this iteration's chunk of the current copy (appended at the bottom) ----
        const bool room = op - fl <= R - 46;                         // this iteration's appends (<= 16 + 11 bytes + 19 of overshoot) stay clear of unflushed output
        const bool near = kind == kK3Near, lit = kind == kK3Lit;
        const bool can = room & (rem > 0) & !((kind == kK3Far) & (gready == 0)) & !(lit & !staged16);
        const int stride = (near & (off < 16)) ? off : 16;          // a near match whose source would overlap the chunk copies `off` bytes and doubles off
        int n = can ? (rem < stride ? rem : stride) : 0;
        n = ((kind == kK3Zero) & (n > 8)) ? 8 : n;
        const int rem_after = rem - n;
        const int op_end = op + rem;                                 // where the current copy ends = where the parsed-ahead sequence's literals go

        // ---- (T4) parse ahead: the next sequence's header (needs only the cursor) ----
        const bool may_parse = (pv == 0) & (final_seen == 0) & !(lit & (rem > 0)) & staged16;
        if (may_parse) {
            const uint32_t tok = hdr ? token : (x0 & 255u);
            const uint32_t t4 = tok >> 4, b1 = (x0 >> 8) & 255u, mlc = tok & 15u;
            const bool e1 = (hdr == 0) & (t4 == 15u), e2 = mlc == 15u;
            const int ll = hdr ? 0 : (int)t4 + (e1 ? (int)b1 : 0);
            const bool in_win = (hdr != 0) | (t4 <= 11u);             // literals (<= 11, no length byte), offset and first match-length byte are within 16 bytes
            const int o = hdr ? 0 : 1 + (int)t4;                      // position of the offset field when in_win
            // offset + first match-length byte, read at their byte position in the staging ring
            const uint32_t fo = (uint32_t)(A + o);
            const uint32_t fa0 = ((fo << 6) & kStageMask & ~0xFFu) | lane4;
            const uint32_t ot = wv::alignbyte(L3_STAGE((fa0 + 256u) & kStageMask), L3_STAGE(fa0), fo & 3u);
            const int vo = (int)(ot & 0xFFFFu);
            const uint32_t extb = (ot >> 16) & 255u;
            const int ml = (int)mlc + kMinMatch + (e2 ? (int)extb : 0);
            const int lit_end = op_end + ll;
            const int p_after = ip + o + 2;                          // after the offset field
            // what the 16-byte view cannot decide goes to the byte-wise parser: the end of the source, length bytes of 255,
            // the final literal run (lz4.c:851 / :965), every error (lz4.c:863,893 / :980,1024)
            bool trap = (ip + 16 > iend) | (e1 & (b1 == 255u)) | (ip + 1 + (e1 ? 1 : 0) + ll > iend);
            trap |= in_win & ((e2 & (extb == 255u)) | (vo > lit_end) | ((int64_t)lit_end + ml > (int64_t)oend - kLastLiterals));
            if (KNOWN) trap |= (hdr == 0) & (lit_end > oend - 8);
            else       trap |= ((hdr == 0) & ((lit_end > oend - kMfLimit) | (ip + 1 + (e1 ? 1 : 0) + ll > iend - 8))) | (in_win & e2 & !(p_after < iend - (kLastLiterals + 1)));
            p_l0 = wv::alignbyte(x1, x0, 1); p_l1 = wv::alignbyte(x2, x1, 1); p_l2 = wv::alignbyte(x3, x2, 1);
            token = tok;
            p_ll = in_win ? ll : 0;
            p_st = in_win ? 0 : ll;
            p_ml = in_win ? ml : 0;
            p_off = vo;
            p_flags = in_win ? 0 : (int)kF3Header;
            const int ip_fast = in_win ? p_after + (e2 ? 1 : 0) : ip + 1 + (e1 ? 1 : 0);
            if (trap) {
                // ---- byte-wise: token + literal length (lz4.c:844 / :957-961), or, in header position, offset + match length
                //      (lz4.c:862-866 / :979-997); literals are always streamed from here, so the header gets its own parse ----
                int err = 0, pos = ip;
                if (!hdr) {
                    const uint32_t tk = ip < iend ? src[ip] : 0u;
                    int l = (int)(tk >> 4);
                    pos = ip + 1;
                    if (l == 15) {
                        uint32_t b = 255;
                        if (KNOWN) { do { b = pos < iend ? src[pos] : 0u; pos++; l += (int)b; if (l > (1 << 30)) { err = -pos; l = 0; break; } } while (b == 255); }
                        else       { while (pos < iend && b == 255) { b = src[pos]; pos++; l += (int)b; l = l > (1 << 30) ? (1 << 30) : l; } }   // saturate: the reference counts in size_t
                    }
                    token = tk;
                    const int le = (int)((int64_t)op_end + l > 0x7FFFFFFF ? 0x7FFFFFFF : op_end + l);
                    const bool last = KNOWN ? (le > oend - 8) : ((le > oend - kMfLimit) | (pos + l > iend - 8));
                    p_ll = 0; p_st = l; p_ml = 0; p_off = 8;
                    if (last) {                                      // final literal run, lz4.c:851-858 / :965-975
                        if (KNOWN) { if (err == 0 && (le != oend || pos + l > iend)) err = -pos; }
                        else       { if (le > oend || pos + l != iend) err = -pos; }
                        p_flags = kF3Final;
                        p_res = KNOWN ? pos + l : le;
                        final_seen = 1;
                    } else {
                        if (KNOWN && err == 0 && pos + l > iend) err = -pos;     // never read literals past the source
                        p_flags = kF3Header;
                    }
                    ip = pos;
                } else {
                    int p = ip;
                    const int o_ = (int)((p < iend ? src[p] : 0u) | ((p + 1 < iend ? src[p + 1] : 0u) << 8));
                    p += 2;
                    int m = (int)(token & 15u);
                    if (m == 15) {
                        if (KNOWN) {
                            uint32_t b;
                            while ((b = (p < iend ? src[p] : 0u)) == 255) { m += 255; p++; if (m > (1 << 30)) { err = -p; break; } }
                            m += (int)b; p++;
                        } else {
                            while (p < iend - (kLastLiterals + 1)) { const uint32_t b = src[p]; p++; m += (int)b; m = m > (1 << 30) ? (1 << 30) : m; if (b != 255) break; }
                        }
                    }
                    m += kMinMatch;
                    if (err != 0) {}
                    else if (op_end - o_ < 0) err = -(ip + 2);
                    else if ((int64_t)op_end + m > (int64_t)oend - kLastLiterals) err = -p;
                    p_ll = 0; p_st = 0; p_ml = m; p_off = o_; p_flags = 0;
                    ip = p;
                    hdr = 0;
                }
                if (err != 0) { p_flags = kF3Err; p_res = err; }
                hdr = (p_flags & kF3Header) ? 1 : 0;
            } else {
                ip = ip_fast;
                hdr = in_win ? 0 : 1;
            }
            pv = 1;
        }

        // ---- (T5) far fetch for the chunk appended at the bottom of the NEXT iteration: ALWAYS one load instruction ----
        // continuation of the current far match, or the first 16 bytes of the parsed-ahead match if the current copy ends
        // in this iteration (the source of the chunk appended at output position p is p - off; a lane that could not append
        // what it holds simply fetches the same bytes again)
        {
            const bool f_cont = (kind == kK3Far) & (rem_after > 0);
            const bool f_first = (rem_after == 0) & room & (pv != 0) & (p_ml != 0) & (p_flags == 0) & (p_off > kNearMax);
            const int f_pos = f_cont ? op + n - off : op_end + p_ll - p_off;
            const bool f_want = f_cont | f_first;
            const bool f_do = f_want & (f_pos + 16 <= fl);
            wv::vm_load16_pred<POL & 3>(f_do, (uint64_t)dst + (uint64_t)(uint32_t)f_pos, ldF);
            flush_blocked = (f_want & !f_do) ? 1 : 0;
            gready = f_do ? 1 : 0;                                   // (only read while kind == kK3Far)
        }

        {
        // ---- (T6) input staging: request new pieces: ALWAYS one load instruction ----
        {
            const int A2 = ip + skew;
            have = ((us_pend == 0) & (A2 >= have)) ? (A2 & ~(PIECE - 1)) : have;   // the cursor is past everything staged (start of the block)
            const int ahead = have - A2;                             // <= PIECE: the older half of the ring is no longer needed
            const bool need = (done == 0) & (us_pend == 0) & (have < in_total) & (ahead <= PIECE);
            const bool urgent = need & (ahead < PIECE - 6);
            const uint64_t needy = wv::ballot(need);
            const int cnt = wv::popc64(needy);
            const bool go = cnt >= (3 * PIECES_PER_LOAD) / 4 || wv::any(urgent);   // wave-uniform
            bool hv = false;
            uint64_t g = 0;
            ld_pend = 0;
            if (go) {
                const int rank = wv::rank_below(needy);
                if (need & (rank < PIECES_PER_LOAD)) {
                    load_rec[rank] = Aligned16{ { (uint32_t)lane, (uint32_t)have, (uint32_t)src_al, (uint32_t)(src_al >> 32) } };
                    ld_pend = 1;
                }
                wv::mem_sync();
                const int idx = lane / HELPERS, sub = lane % HELPERS;
                hv = idx < (cnt < PIECES_PER_LOAD ? cnt : PIECES_PER_LOAD);
                if (hv) {
                    const Aligned16 r = load_rec[idx];
                    g = ((uint64_t)r.w[2] | ((uint64_t)r.w[3] << 32)) + (uint64_t)r.w[1] + (uint64_t)(16 * sub);
                    ld_hdst = ((((r.w[1] >> 2) + 4u * (uint32_t)sub) << 8) & kStageMask) | (r.w[0] << 2);   // first of my four staging rows
                }
                wv::mem_sync();
            }
            ld_hvalid = hv ? 1 : 0;
            ld_inflight = go;
            wv::vm_load16_pred<(POL >> 2) & 3>(hv, g, ldH);
        }
        }

        // ================================ BOTTOM ================================
        // ---- (B1) the loads of the PREVIOUS iteration have landed (this iteration's four accesses stay in flight) ----
        wv::vm_wait<4>(usF, usH);

        // ---- (B2) land the previous iteration's pieces ----
        if (us_inflight) {                                         // wave-uniform
            if (us_hvalid) {
                uint32_t* d = (uint32_t*)(stage + us_hdst);          // four consecutive rows of the piece: no wrap inside
                d[0] = usH.x; d[64] = usH.y; d[128] = usH.z; d[192] = usH.w;
            }
            wv::mem_sync();
            have += us_pend ? PIECE : 0;
            us_pend = 0;
            us_inflight = false;
        }

        // ---- (B3) append the chunk ----
        {
            const bool far_src = kind == kK3Far;
            v0 = lit ? x0 : (far_src ? usF.x : v0);
            v1 = lit ? x1 : (far_src ? usF.y : v1);
            v2 = lit ? x2 : (far_src ? usF.z : v2);
            v3 = lit ? x3 : (far_src ? usF.w : v3);
            if ((kind == kK3Zero) & (n > 0)) {                       // offset 0 (corrupt streams only): keep what dst holds, 8 bytes at a time
                uint64_t acc = 0;
                for (int b = 0; b < n; b++) acc |= (uint64_t)dst[op + b] << (8 * b);
                v0 = (uint32_t)acc; v1 = (uint32_t)(acc >> 32);
            }
            ip += lit ? n : 0;
            L3_APPEND(v0, v1, v2, v3, n, true);
            rem = rem_after;
            off = (near & (off < 16) & (n > 0)) ? off * 2 : off;
            kind = rem == 0 ? (int)kK3None : kind;
        }

        // ---- (B4) promote the parsed-ahead sequence: its inline literals, then its copy becomes the current one ----
        {
            const bool promote = (rem == 0) & (pv != 0) & room;
            const bool perr = promote & ((p_flags & kF3Err) != 0);
            const bool pgo = promote & !perr;
            L3_APPEND(p_l0, p_l1, p_l2, 0u, pgo ? p_ll : 0, false);
            if (perr) {                                              // corrupt stream: this lane is finished, nothing more is stored
                done = 1; final_seen = 1; final_run = 0; result = p_res;
            }
            if (pgo) {
                const bool streamed = p_st > 0;
                rem = streamed ? p_st : p_ml;
                off = streamed ? off : p_off;
                kind = streamed ? (int)kK3Lit : (p_ml == 0 ? (int)kK3None : (p_off == 0 ? (int)kK3Zero : (p_off <= kNearMax ? (int)kK3Near : (int)kK3Far)));
                final_run = (p_flags & kF3Final) ? 1 : final_run;
                result = (p_flags & kF3Final) ? p_res : result;
            }
            pv = promote ? 0 : pv;
        }

        // ---- (B5) end of block: write out the last bytes exactly ----
        if (final_run && rem == 0 && !pv && !done) {
            uint32_t qa = ring_add(oa, kRingBytes - ((((uint32_t)(op - fl)) << 6) & ~0xFFu));
            while (op - fl >= 4) { const uint32_t qd = L3_RING(qa); __builtin_memcpy(dst + fl, &qd, 4); fl += 4; qa = ring_add(qa, 256u); }
            if (fl < op) {
                const uint32_t qd = L3_RING(qa);
                for (int b = 0; fl + b < op; b++) dst[fl + b] = (uint8_t)(qd >> (8 * b));
            }
            done = 1;
        }
        return !wv::any(done == 0);                                  // every lane of the wavefront is finished
    };

    for (;;) {
        if (iteration(fa, fb, ha, h_dst_a, h_valid_a, inflight_a, pend_a, hb, h_dst_b, h_valid_b, inflight_b, pend_b)) break;
        if (iteration(fb, fa, hb, h_dst_b, h_valid_b, inflight_b, pend_b, ha, h_dst_a, h_valid_a, inflight_a, pend_a)) break;
    }
    return result;
#undef L3_RING
#undef L3_STAGE
#undef L3_PHASE_SEL
#undef L3_APPEND
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.
template <bool KNOWN, int R, int SB, int POL = 0>
__global__ void __launch_bounds__(64) decode_lane3_kernel(Batch b, int filter)
{
    LZ4HIP_STATIC_LDS(lds, lane3_lds_bytes(R, SB));
    const int lane = (int)threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x * 64 + lane;
    bool active = blk < b.n_blocks;
    int src_len = 0, out_size = 0;
    if (active) {
        src_len = batch_src_len(b, blk); out_size = batch_dst_cap(b, blk);
        active = block_selected(filter, src_len, out_size);
    }
    if (!wv::any(active)) return;
    const uint8_t* src = active ? batch_src(b, blk) : nullptr;
    uint8_t* dst = active ? batch_dst(b, blk) : nullptr;
    const int r = lane3_decode_block<KNOWN, R, SB, POL>(lds, lane, active, src, src_len, dst, out_size);
    if (active) b.result[blk] = r;
}

}  // namespace lz4hip
