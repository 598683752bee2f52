// (NOT part of the product library: the second generation of the lane decoder, superseded by lz4net_amd/csrc/lz4hip_decode_lane4.hpp;
//  kept under tools/ab/ for A/B runs -- libraries built with -DLZ4HIP_TUNING_BUILD -- and emulator tests)
// lz4hip_decode_lane.hpp -- lane-per-block LZ4 decoder: 64 blocks per wavefront, every byte of global traffic moved by
// wave-cooperative accesses, and exactly one vector-memory wait per loop iteration, for loads issued a full
// iteration earlier.
// Same functions / return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).
//
// Every loop iteration every lane (1) reads the 16 bytes at its input cursor, (3) parses the NEXT sequence header
// if none is waiting, (5) appends at most 16 bytes of its current copy to its output ring and, when that copy
// is finished, (6) appends the waiting sequence's literals and starts its match, (7) takes part in the cooperative
// flush.  What shapes it (counters in profiles/r01 and profiles/r02):
//   * INPUT through LDS.  The round-1 kernel re-fetched every 64-byte line of every lane's compressed stream four
//     times (16-byte per-lane window loads; nothing survives in a cache with a million blocks in flight) and ran
//     into the rate at which the fabric serves sector requests (~67 G/s).  Here a lane's stream is staged in a
//     64-byte ring in LDS, filled in aligned 32-byte pieces by cooperative loads (two lanes per piece, up to 32
//     pieces per load instruction): each byte of input crosses the fabric once.  Header bytes, inline literals
//     and streamed literal runs all come out of that ring (5 aligned ds_read_b32 + 4 v_perm_b32 per iteration).
//   * OUTPUT ring per lane in LDS (128 bytes, dword-interleaved across the lanes, byte-granular appends through
//     v_perm_b32 on aligned dwords), finished 64-byte lines leave four lanes per line.
//   * LDS per wavefront decides the residency, and throughput follows residency (profiles/r02/decoder_ab_*.txt:
//     6 / 8 / 9 / 12 wavefronts per CU -> 436 / 500 / 542 / 628 GB/s on the same code), so both rings are as small
//     as they can be: 64 x (128 + 64) + 1 KiB = 13.3 KiB -> 12 wavefronts per CU.
//   * PARSE AHEAD.  The header of sequence k+1 is parsed before the copy of sequence k is finished (it only needs
//     the cursor); its literals wait in registers.  If its match lies behind the ring, the fetch of the first 16
//     source bytes leaves in the iteration that parsed it -- the address follows from arithmetic on lengths -- and
//     is consumed in the next one.  Far-match data alternates between two register sets by iteration parity (the
//     loop body is instantiated twice) and the pieces of the input staging are written to LDS one iteration after
//     they were requested; all loads of an iteration are issued AFTER the iteration has read last iteration's data,
//     so the loop has a single s_waitcnt vmcnt(0) and it waits for loads that are one full iteration old
//     (wave-cycles parked at s_waitcnt: 67 % in round 1, 38 % now).
//   * Matches with an offset below 16 copy `offset` bytes and then DOUBLE the offset (what has been produced is
//     periodic): 1, 2, 4, 8, 16 bytes per iteration without any special periodic-pattern machinery.
//   * Boolean state is combined with the bitwise operators on purpose: the short-circuit forms compile to
//     exec-mask branches, the bitwise ones to scalar mask arithmetic next to the vector ALU's work.
#pragma once
#include "lz4hip_common.hpp"

#ifndef LZ4HIP_ITERATION_HOOK
#define LZ4HIP_ITERATION_HOOK(lane) ((void)0)    /* the emulator build counts loop iterations here (tests/simt) */
#endif

namespace lz4hip {

#ifndef LZ4HIP_DEC_FLUSH_RECS
#define LZ4HIP_DEC_FLUSH_RECS 32      /* the emulator also builds a 'starved' variant with 4 (tests/simt/build_emu.py) */
#endif
constexpr int kDecFlushRecs = LZ4HIP_DEC_FLUSH_RECS;          // flush records (lines stored per flush round, two store instructions)
constexpr int kDecFlushRecBytes = 16 * kDecFlushRecs;
constexpr int kDecLoadRecBytes = 512;                     // up to 32 piece-load records
// LDS of one wavefront: 64 output rings, 64 input staging rings (both dword-interleaved across the lanes), the
// records of the cooperative stores and loads.  What is left of 160 KiB decides how many
// wavefronts a CU holds, and the decoder's speed is proportional to that (profiles/r02).
constexpr unsigned lane_decode_lds_bytes(int ring_bytes, int stage_bytes)
{
    return 64u * (unsigned)(ring_bytes + stage_bytes) + (unsigned)(kDecFlushRecBytes + kDecLoadRecBytes);
}

// what the next chunk of a lane's current copy is made from
enum LaneMode { kLIdle = 0, kLNear = 2, kLGlobal = 3, kLLit = 4, kLZeroOff = 5 };
enum LanePending { kLNeedToken = 0, kLNeedHeader = 1 };

// All 64 lanes of the wavefront call this together and stay in the loop until the last one is done: a lane without
// a block (`active` false) or with a finished block still lends a hand to the cooperative loads and stores.
template <bool KNOWN, int R, int SB>
LZ4HIP_DEVICE int lane_decode_block(unsigned char* lds, int lane, bool active, const uint8_t* __restrict__ src, int iend,
                                    uint8_t* dst, int oend)
{
    constexpr int RW = R / 4;                                        // ring dwords per lane
    static_assert(R >= 128 && (R & (R - 1)) == 0, "ring: power of two, >= 128 bytes");
    static_assert(SB == 64 || SB == 128, "staging ring: 64 or 128 bytes per lane");
    constexpr int PIECE = SB / 2;                                    // the input arrives in aligned pieces of half a staging ring
    constexpr int HELPERS = PIECE / 16;                              // lanes that load one piece (16 bytes each)
#ifdef LZ4HIP_DEC_LOAD_PIECES                                        /* the emulator's 'starved' build: 2 pieces per round */
    constexpr int PIECES_PER_LOAD = LZ4HIP_DEC_LOAD_PIECES;
#else
    constexpr int PIECES_PER_LOAD = 64 / HELPERS;                    // pieces one load instruction brings in
#endif
    constexpr uint32_t kStageMask = (uint32_t)(SB / 4 - 1) << 8;
    // An append writes whole dwords, up to 19 bytes past its last byte; those land on ring bytes op-R+19 and older.
    constexpr int kNearMax = R - 20;                                 // largest offset served from the ring
    // unflushed output must survive the appends of the next iteration (<= 16 + 11 bytes) and must not reach back
    // further than the nearest far source (offset > kNearMax, 16 bytes fetched)
    constexpr int kFlushUrgent = R >= 256 ? 128 : 64;
    const uint32_t lane4 = (uint32_t)lane << 2;
    constexpr uint32_t kRingMask = (uint32_t)(RW - 1) << 8;
    unsigned char* stage = lds + 64 * R;
    Aligned16* flush_rec = (Aligned16*)(stage + 64 * SB);
    Aligned16* load_rec = (Aligned16*)(stage + 64 * SB + kDecFlushRecBytes);
    // dword k of this lane's ring lives at LDS byte ((k & (RW-1)) << 8) | (lane << 2); RING_AT(p, j) is the dword j
    // dwords after the one containing output byte p
#define RING_AT(p, j) (*(uint32_t*)(lds + ((((((uint32_t)(p)) << 6) + 256u * (uint32_t)(j)) & kRingMask) | lane4)))
    // staging: dword k of this lane's SB-byte ring at stage + ((k & (SB/4-1)) << 8) | (lane << 2); STAGE_AT(a, j) is the
    // dword j dwords after the one containing stream byte a (aligned coordinates)
#define STAGE_AT(a, j) (*(const uint32_t*)(stage + ((((((uint32_t)(a)) << 6) + 256u * (uint32_t)(j)) & kStageMask) | lane4)))

    // ---- per-lane state (plain integers: bools would live in SGPR lane masks) ----
    // input: positions are block coordinates; `skew` converts to the aligned coordinates the staging works in
    const int skew = (int)((uint64_t)src & (uint64_t)(PIECE - 1));
    const uint64_t src_al = (uint64_t)src - (uint64_t)skew;
    const int in_total = iend > 0 ? (int)(((int64_t)skew + iend + PIECE - 1) & ~(int64_t)(PIECE - 1)) : 0;
    int ip = 0;                  // input cursor: next header, or next literal of a streamed literal run
    int in_have = 0;             // aligned coordinates: bytes [in_have - SB, in_have) are staged (where they exist)
    int in_pending = 0;          // a line load of this lane is in flight
    // output
    int op = 0, flushed = 0;     // bytes produced / bytes already stored to dst (multiple of 64)
    uint32_t tail = 0;           // ring dword containing op: its low (op & 3) bytes are output, the rest is junk
    // current copy
    int mode = kLIdle, rem = 0;
    int off = 8;                 // offset of the current match; a near match with off < 16 DOUBLES it after every chunk (what has been
                                 // produced is periodic with period off, hence with period 2*off): 1, 2, 4, 8, then 16 bytes per iteration
    uint32_t fa0 = 0, fa1 = 0, fa2 = 0, fa3 = 0, fb0 = 0, fb1 = 0, fb2 = 0, fb3 = 0;   // far-match data, by iteration parity
    int gready = 0;              // kLGlobal: the 16 bytes fetched in the previous iteration are this lane's next chunk
    // parsed-ahead sequence
    int nx = 0;                  // there is one
    int n_ll = 0, n_stream = 0, n_ml = 0, n_off = 8, n_hasmatch = 0, n_final = 0, n_result = 0, n_err = 0;
    uint32_t n_l0 = 0, n_l1 = 0, n_l2 = 0;     // its inline literals
    int pend = kLNeedToken;
    uint32_t token = 0;
    int final_seen = 0;          // the last sequence has been parsed
    int final_run = 0, result = 0;
    int done = 0;
    int flush_blocked = 0;       // last iteration's far fetch had to wait for bytes that are not in global memory yet
    if (!active || (!KNOWN && iend == 0)) { done = 1; final_seen = 1; }   // lz4.c:946 returns -(0)
    // cooperative line loads in flight (helper role): data, destination in the staging area
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t h_dst = 0;
    int h_valid = 0;
    bool inflight = false;       // wave-uniform

    // Append the low n_ bytes of the data dwords to the output ring at op: data
    // rotated to the byte phase of op with one v_perm_b32 per dword, whole dwords stored, first dword merged with
    // the bytes below op.  The new tail is picked from the dwords just written (no read-back).
#define APPEND_HEAD()                                                                                   \
        const uint32_t sb_ = (uint32_t)op & 3u;                                                         \
        const uint32_t s_ = wv::alignbyte(0x08070605u, 0x04030201u, sb_ ^ 3u);                          \
        const uint32_t keep_ = (1u << (8u * sb_)) - 1u;                                                 \
        const int k_ = op
#define APPEND4(d0_, d1_, d2_, d3_, n_)                                                                 \
    do {                                                                                                \
        APPEND_HEAD();                                                                                  \
        const uint32_t a0_ = (tail & keep_) | wv::perm(d0_, 0u, s_);                                    \
        const uint32_t a1_ = wv::perm(d1_, d0_, s_), a2_ = wv::perm(d2_, d1_, s_);                      \
        const uint32_t a3_ = wv::perm(d3_, d2_, s_), a4_ = wv::perm(0u, d3_, s_);                       \
        RING_AT(k_, 0) = a0_; RING_AT(k_, 1) = a1_; RING_AT(k_, 2) = a2_; RING_AT(k_, 3) = a3_; RING_AT(k_, 4) = a4_; \
        op += (n_);                                                                                     \
        tail = RING_AT(op, 0);                                                                          \
    } while (0)
#define APPEND3(d0_, d1_, d2_, n_)                                                                      \
    do {                                                                                                \
        APPEND_HEAD();                                                                                  \
        const uint32_t a0_ = (tail & keep_) | wv::perm(d0_, 0u, s_);                                    \
        const uint32_t a1_ = wv::perm(d1_, d0_, s_), a2_ = wv::perm(d2_, d1_, s_), a3_ = wv::perm(0u, d2_, s_); \
        RING_AT(k_, 0) = a0_; RING_AT(k_, 1) = a1_; RING_AT(k_, 2) = a2_; RING_AT(k_, 3) = a3_;         \
        op += (n_);                                                                                     \
        tail = RING_AT(op, 0);                                                                          \
    } while (0)
    // selector that extracts 4 bytes at byte phase (p & 3) from a dword pair: wv::perm(hi, lo, PHASE_SEL(p))
#define PHASE_SEL(p_) wv::alignbyte(0x07060504u, 0x03020100u, (uint32_t)(p_) & 3u)

    // One iteration.  ld*: the far-match registers loaded in THIS iteration; us*: those loaded in the previous one.
    auto iteration = [&](uint32_t& ld0, uint32_t& ld1, uint32_t& ld2, uint32_t& ld3,
                         const uint32_t& us0, const uint32_t& us1, const uint32_t& us2, const uint32_t& us3) __attribute__((always_inline)) -> bool {
        LZ4HIP_ITERATION_HOOK(lane);
        // =========================== (1) the 16 bytes at the input cursor ===========================
        const int A = ip + skew;
        const uint32_t z0 = STAGE_AT(A, 0), z1 = STAGE_AT(A, 1), z2 = STAGE_AT(A, 2), z3 = STAGE_AT(A, 3), z4 = STAGE_AT(A, 4);
        const uint32_t sx = PHASE_SEL(A);
        const uint32_t x0 = wv::perm(z1, z0, sx), x1 = wv::perm(z2, z1, sx), x2 = wv::perm(z3, z2, sx), x3 = wv::perm(z4, z3, sx);
        const bool x_ok = ip + 16 <= iend;                           // the 16 bytes are inside the source ...
        const bool x_have = in_have - A >= 16;                       // ... and staged

        const bool lit_slow = (mode == kLLit) & !x_ok;               // the last bytes of the source: byte-wise
        const int op_end = op + rem;                                 // where the current copy ends = where the parsed-ahead sequence starts

        // =========================== (3) parse ahead ===========================
        const bool cursor_busy = (mode == kLLit) & (rem > 0);
        const bool may_parse = (nx == 0) & (final_seen == 0) & !cursor_busy & (!x_ok | x_have);
        if (may_parse) {
            // ---- token [+ one literal-length byte] [+ <= 11 literals] + offset [+ one match-length byte]; after a literal
            //      run that did not fit (pend == kLNeedHeader) the same code parses just "offset + match length" ----
            int err = 0;
            const bool hdr = pend == kLNeedHeader;
            uint32_t tok = hdr ? token : (x0 & 255u);
            const uint32_t b1 = (x0 >> 8) & 255u;
            const bool ext1 = !hdr & ((tok >> 4) == 15u);
            int ll = hdr ? 0 : (int)(tok >> 4) + (ext1 ? (int)b1 : 0);
            int pos = ip + (hdr ? 0 : 1 + (ext1 ? 1 : 0));           // position after token (+ literal-length byte)
            if (!hdr & (!x_ok | (ext1 & (b1 == 255u)))) {           // rare: byte-wise, lz4.c:844 / :957-961
                tok = ip < iend ? src[ip] : 0u;
                ll = (int)(tok >> 4);
                pos = ip + 1;
                if (ll == 15) {
                    uint32_t b = 255;
                    if (KNOWN) { do { b = pos < iend ? src[pos] : 0u; pos++; ll += (int)b; if (ll > (1 << 30)) { err = -pos; ll = 0; break; } } while (b == 255); }
                    else       { while (pos < iend && b == 255) { b = src[pos]; pos++; ll += (int)b; ll = ll > (1 << 30) ? (1 << 30) : ll; } }   // saturate: the reference counts in size_t
                }
            }
            token = tok;
            const uint32_t mlc = tok & 15u;
            const bool in_win = x_ok & (ll <= 11);                    // literals, offset and first match-length byte are in x0..x3
            const int lit_end = (int)((int64_t)op_end + ll > 0x7FFFFFFF ? 0x7FFFFFFF : op_end + ll);
            const bool last = !hdr & (KNOWN ? (lit_end > oend - 8) : ((lit_end > oend - kMfLimit) | (pos + ll > iend - 8)));
            n_l0 = wv::alignbyte(x1, x0, 1); n_l1 = wv::alignbyte(x2, x1, 1); n_l2 = wv::alignbyte(x3, x2, 1);
            n_ll = in_win ? ll : 0;
            n_stream = in_win ? 0 : ll;
            n_hasmatch = 0; n_final = 0;
            if (last) {                                              // final literal run, lz4.c:851-858 / :965-975
                if (KNOWN) { if (err == 0 && (lit_end != oend || pos + ll > iend)) err = -pos; }
                else       { if (lit_end > oend || pos + ll != iend) err = -pos; }
                n_final = 1;
                n_result = KNOWN ? pos + ll : lit_end;
                final_seen = 1;
                ip = in_win ? pos + ll : pos;
            } else {
                err = (KNOWN & !hdr & (err == 0) & (pos + ll > iend)) ? -pos : err;   // never read literals past the source
                const int o = hdr ? 0 : 1 + ll, oq = o >> 2;         // offset at bytes o, o+1 of x (o <= 12 when in_win), length byte at o+2
                const uint32_t xl = oq < 2 ? (oq == 0 ? x0 : x1) : (oq == 2 ? x2 : x3);
                const uint32_t xh = oq < 2 ? (oq == 0 ? x1 : x2) : x3;
                const uint32_t ot = wv::alignbyte(xh, xl, (uint32_t)o & 3u);   // bytes o .. o+3
                const uint32_t vo = ot & 0xFFFFu, extb = (ot >> 16) & 255u;
                const int p_off = ip + o + 2;                        // after the offset
                // (the unknown-size decoder only reads a match-length byte while p < iend - 6, lz4.c:986)
                const bool fast = in_win & ((mlc != 15u) | ((extb != 255u) & (KNOWN | (p_off < iend - (kLastLiterals + 1)))));
                const int ml_fast = (int)mlc + kMinMatch + (mlc == 15u ? (int)extb : 0);
                const int ip_fast = p_off + (mlc == 15u ? 1 : 0);
                int ip_next = fast ? ip_fast : (in_win ? pos + ll : pos);
                if (fast) {
                    n_off = (int)vo;
                    n_ml = ml_fast;
                    n_hasmatch = 1;
                    err = ((err == 0) & (lit_end - n_off < 0)) ? -p_off : err;                          // lz4.c:863 / :980
                    err = ((err == 0) & ((int64_t)lit_end + n_ml > (int64_t)oend - kLastLiterals)) ? -ip_fast : err;   // lz4.c:893 / :1024
                } else if (hdr) {                                    // rare: byte-wise, lz4.c:862-866 / :979-997
                    int p = ip;
                    const int o_ = (int)((p < iend ? src[p] : 0u) | ((p + 1 < iend ? src[p + 1] : 0u) << 8));
                    p += 2;
                    int ml = (int)mlc;
                    if (ml == 15) {
                        if (KNOWN) {
                            uint32_t b;
                            while ((b = (p < iend ? src[p] : 0u)) == 255) { ml += 255; p++; if (ml > (1 << 30)) { err = -p; break; } }
                            ml += (int)b; p++;
                        } else {
                            while (p < iend - (kLastLiterals + 1)) { const uint32_t b = src[p]; p++; ml += (int)b; ml = ml > (1 << 30) ? (1 << 30) : ml; if (b != 255) break; }
                        }
                    }
                    ml += kMinMatch;
                    if (err != 0) {}
                    else if (op_end - o_ < 0) err = -(ip + 2);
                    else if ((int64_t)op_end + ml > (int64_t)oend - kLastLiterals) err = -p;
                    n_off = o_; n_ml = ml; n_hasmatch = 1;
                    ip_next = p;
                }
                pend = (fast | hdr) ? (int)kLNeedToken : (int)kLNeedHeader;
                ip = ip_next;
            }
            n_err = err;
            nx = 1;
        }

        // =========================== (5a) the source bytes of this iteration's chunk ===========================
        // (read BEFORE this iteration's loads are issued: the far-match registers `us` were loaded one full iteration
        //  ago, so the only vector-memory wait of the loop -- here -- finds them finished)
        const bool near = mode == kLNear;
        uint32_t v0, v1, v2, v3;
        {
            const int sp = op - off;
            uint32_t r0 = RING_AT(sp, 0), r1 = RING_AT(sp, 1), r2 = RING_AT(sp, 2), r3 = RING_AT(sp, 3), r4 = RING_AT(sp, 4);
            const uint32_t sr = PHASE_SEL(sp);
            const bool lit = mode == kLLit;
            v0 = near ? wv::perm(r1, r0, sr) : (lit ? x0 : us0);
            v1 = near ? wv::perm(r2, r1, sr) : (lit ? x1 : us1);
            v2 = near ? wv::perm(r3, r2, sr) : (lit ? x2 : us2);
            v3 = near ? wv::perm(r4, r3, sr) : (lit ? x3 : us3);
            LZ4HIP_KEEP(v0); LZ4HIP_KEEP(v1); LZ4HIP_KEEP(v2); LZ4HIP_KEEP(v3);
        }

        // =========================== (7) flush finished output, 64 bytes at a time, four lanes per line ===========================
        // (placed right after the iteration's only vector-memory wait: on gfx9 vmcnt counts stores as well, so stores issued just
        //  BEFORE the wait -- at the end of the previous iteration, where this block used to be -- made every iteration that
        //  followed a flush wait for the write acknowledgements; here they have a whole iteration)
        {
            const bool need = (done == 0) & (op - flushed >= 64);
            const bool urgent = need & ((op - flushed >= kFlushUrgent) | (flush_blocked != 0) | ((final_run != 0) & (rem == 0)));
            const uint64_t needy = wv::ballot(need);
            const int cnt_all = wv::popc64(needy);
            if (cnt_all >= 16 || wv::any(urgent)) {                  // wave-uniform
                const int cnt = cnt_all < kDecFlushRecs ? cnt_all : kDecFlushRecs;
                const int frank = wv::rank_below(needy);
                const bool mine = need & (frank < kDecFlushRecs);     // (the others come next iteration)
                if (mine) {
                    const uint64_t dp = (uint64_t)dst;
                    flush_rec[frank] = Aligned16{ { (uint32_t)lane, (uint32_t)flushed, (uint32_t)dp, (uint32_t)(dp >> 32) } };
                }
                wv::mem_sync();
                const int sub = lane & 3;
                for (int base = 0; base < cnt; base += 16) {         // wave-uniform trip count
                    const int idx = base + (lane >> 2);
                    if (idx < cnt) {
                        const Aligned16 r = flush_rec[idx];
                        const int fj = (int)r.w[1];
                        const uint64_t dj = (uint64_t)r.w[2] | ((uint64_t)r.w[3] << 32);
                        // 16 dwords of lane r.w[0]'s ring from fj (a multiple of 64: no wrap), this lane takes 4 of them
                        const uint32_t* fp = (const uint32_t*)(lds + ((((uint32_t)fj << 6) & kRingMask) | (r.w[0] << 2))) + 4 * sub * 64;
                        wv::store_global16(dj + (uint64_t)(fj + 16 * sub), fp[0], fp[64], fp[128], fp[192]);
                    }
                }
                wv::mem_sync();                                      // records and ring bytes are free to be overwritten again
                flushed += mine ? 64 : 0;
            }
        }

        // =========================== (2) size of this iteration's chunk of the current copy ===========================
        // this iteration's appends (<= 16 + 11 bytes, written with up to 19 bytes of overshoot) must not reach unflushed output
        const bool room = op - flushed <= R - 46;
        const bool can = room & (rem > 0) & !((mode == kLGlobal) & (gready == 0)) & !((mode == kLLit) & x_ok & !x_have);
        const int stride = (near & (off < 16)) ? off : 16;          // bytes per chunk: at most `off` while a near match's source would overlap the chunk
        int n = can ? (rem < stride ? rem : stride) : 0;
        const bool slow8 = can & (lit_slow | (mode == kLZeroOff));    // (branch-free: bitwise operators on purpose, the
        n = (slow8 & (n > 8)) ? 8 : n;                               //  short-circuit forms compile to exec-mask branches)
        const int rem_after = rem - n;
        if (can && (lit_slow || mode == kLZeroOff)) {                 // rare byte-wise sources, 8 bytes at a time
            uint64_t acc = 0;
            if (lit_slow) { for (int b = 0; b < n; b++) if (ip + b < iend) acc |= (uint64_t)src[ip + b] << (8 * b); }
            else          { for (int b = 0; b < n; b++) acc |= (uint64_t)dst[op + b] << (8 * b); }   // offset 0: keep what dst holds
            v0 = (uint32_t)acc; v1 = (uint32_t)(acc >> 32);
        }

        // =========================== (4a) input staging: land last iteration's pieces, request new ones ===========================
        // (every load of an iteration is issued here and in (4b), after (5a), and consumed one full iteration later)
        if (inflight) {                                              // wave-uniform
            if (h_valid) {
                uint32_t* d = (uint32_t*)(stage + h_dst);            // four consecutive dwords of the piece: no wrap inside
                d[0] = h0; d[64] = h1; d[128] = h2; d[192] = h3;
            }
            wv::mem_sync();
            in_have += in_pending ? PIECE : 0;
            in_pending = 0;
            inflight = false;
        }
        {
            const int A2 = ip + skew;
            // cursor beyond everything staged (start of the block, or a rare path jumped ahead): restart at its line
            in_have = ((in_pending == 0) & (A2 >= in_have)) ? (A2 & ~(PIECE - 1)) : in_have;
            const int ahead = in_have - A2;                          // <= PIECE: the older half of the staging ring is no longer needed
            const bool need = (done == 0) & (in_pending == 0) & (in_have < in_total) & (ahead <= PIECE);
            const bool urgent = need & (ahead < PIECE - 6);           // (a short sequence consumes ~5 bytes per iteration, a load takes two)
            const uint64_t needy = wv::ballot(need);
            const int cnt = wv::popc64(needy);
            if (cnt >= (3 * PIECES_PER_LOAD) / 4 || wv::any(urgent)) {   // wave-uniform
                const int rank = wv::rank_below(needy);
                if (need & (rank < PIECES_PER_LOAD)) {
                    load_rec[rank] = Aligned16{ { (uint32_t)lane, (uint32_t)in_have, (uint32_t)src_al, (uint32_t)(src_al >> 32) } };
                    in_pending = 1;
                }
                wv::mem_sync();
                const int idx = lane / HELPERS, sub = lane % HELPERS;
                h_valid = idx < (cnt < PIECES_PER_LOAD ? cnt : PIECES_PER_LOAD);
                if (h_valid) {
                    const Aligned16 r = load_rec[idx];
                    const uint64_t g = ((uint64_t)r.w[2] | ((uint64_t)r.w[3] << 32)) + (uint64_t)r.w[1] + (uint64_t)(16 * sub);
                    wv::load_global16(g, h0, h1, h2, h3);
                    const uint32_t dw = ((r.w[1] >> 2) + 4u * (uint32_t)sub) & (uint32_t)(SB / 4 - 1);   // first of my four staging dwords
                    h_dst = (dw << 8) | (r.w[0] << 2);
                }
                wv::mem_sync();
                inflight = true;
            }
        }

        // =========================== (4b) far fetch for the NEXT iteration's chunk ===========================
        // continuation of the current far match, or the first 16 bytes of the parsed-ahead match if the current
        // copy ends in this iteration (then the sequence is promoted below and its match starts next iteration)
        const bool f_cont = (mode == kLGlobal) & (rem_after > 0);
        const bool f_first = (rem_after == 0) & room & (nx != 0) & (n_hasmatch != 0) & (n_err == 0) & (n_off > kNearMax);
        // (the source of the chunk that will be appended at output position p is p - off: no separate fetch cursor, so a lane
        //  that holds fetched bytes but could not append them -- it missed two flush rounds in a row, its ring is full --
        //  simply fetches the same 16 bytes again; tests/test_simt_emulation.py::test_lane_decoder_starved_flush)
        const int f_pos = f_cont ? op + n - off : op_end + n_ll - n_off;
        const bool f_want = f_cont | f_first;
        const bool f_do = f_want & (f_pos + 16 <= flushed);
        if (f_do) {
            const Vec16 w = load_v16(dst + f_pos);
            ld0 = w.w[0]; ld1 = w.w[1]; ld2 = w.w[2]; ld3 = w.w[3];
        }
        flush_blocked = (f_want & !f_do) ? 1 : 0;                    // (read by the flush of the NEXT iteration)

        // =========================== (5b) append the chunk ===========================
        {
            ip += mode == kLLit ? n : 0;
            APPEND4(v0, v1, v2, v3, n);
            rem = rem_after;
            const bool grow = near & (off < 16) & (n > 0);
            off = grow ? off * 2 : off;
            mode = rem == 0 ? (int)kLIdle : mode;
        }

        // =========================== (6) promote the parsed-ahead sequence ===========================
        const bool promote = (rem == 0) & (nx != 0) & room;
        const bool perr = promote & (n_err != 0);
        const bool pgo = promote & (n_err == 0);
        APPEND3(n_l0, n_l1, n_l2, pgo ? n_ll : 0);
        if (perr) {                                                  // corrupt stream: this lane is finished, nothing more is stored
            done = 1; final_seen = 1; final_run = 0; result = n_err;
        }
        if (pgo) {
            rem = n_stream;
            mode = n_stream > 0 ? (int)kLLit : (int)kLIdle;
            final_run = n_final;
            result = n_final ? n_result : result;
        }
        if (pgo & (n_hasmatch != 0)) {
            // ---- start the match copy: byte-wise semantics out[i] = out[i - off] ----
            off = n_off;
            mode = off == 0 ? (int)kLZeroOff : (off <= kNearMax ? (int)kLNear : (int)kLGlobal);
            rem = n_ml;
        }
        nx = promote ? 0 : nx;
        gready = ((mode == kLGlobal) & f_do) ? 1 : 0;

        if (final_run && rem == 0 && !nx && !done) {
            // ---- end of block: write out the last bytes exactly ----
            while (op - flushed >= 4) { const uint32_t qd = RING_AT(flushed, 0); __builtin_memcpy(dst + flushed, &qd, 4); flushed += 4; }
            if (flushed < op) {
                const uint32_t qd = (flushed >> 2) == (op >> 2) ? tail : RING_AT(flushed, 0);
                for (int b = 0; flushed + b < op; b++) dst[flushed + b] = (uint8_t)(qd >> (8 * b));
            }
            done = 1;
        }
        return !wv::any(done == 0);                                  // every lane of the wavefront is finished
    };

    for (;;) {
        if (iteration(fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3)) break;
        if (iteration(fb0, fb1, fb2, fb3, fa0, fa1, fa2, fa3)) break;
    }
    return result;
#undef RING_AT
#undef STAGE_AT
#undef APPEND_HEAD
#undef APPEND4
#undef APPEND3
#undef PHASE_SEL
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.
template <bool KNOWN, int R, int SB>
__global__ void __launch_bounds__(64) decode_lane_kernel(Batch b, int filter)
{
    LZ4HIP_STATIC_LDS(lds, lane_decode_lds_bytes(R, SB));
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
    const int r = lane_decode_block<KNOWN, R, SB>(lds, lane, active, src, src_len, dst, out_size);
    if (active) b.result[blk] = r;
}

}  // namespace lz4hip
