// lz4hip_decode_chunked.hpp -- lane-per-block LZ4 decoder as a CONVERGENT state machine:
// every loop iteration every lane (b) produces at most 16 output bytes from whatever source its state
// says, (a) parses a sequence header if its copy is finished -- appending the sequence's literals right
// away when they sit in the header window --, (c) takes part in the cooperative flush of finished 64-byte
// lines, (d) requests the next 16 (or 32) source bytes if it is copying from global memory.
// Same functions / return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).
//
// Why this shape (measured on MI355X, profiles/r01): with one lane per block the cost is not
// arithmetic but (1) vector-memory instructions -- a wavefront instruction whose 64 lanes touch 64
// different lines costs 3..30 CU-cycles PER LANE in the texture-address/L1 pipeline, (2) divergence --
// a sequence-per-iteration loop executes the union of all lanes' paths (~1300 instructions and ~40
// vector-memory instructions per iteration in lz4hip_decode_staged.hpp), (3) any load that is
// consumed in the iteration that issues it stalls the whole wavefront for a memory round trip, and
// with 64 lanes "some lane needs one" is true every iteration, and (4) exec-mask bookkeeping: the
// first version of this kernel spent ~600 of its ~1400 instructions on s_and_saveexec / s_or / s_andn2
// for nested ifs and bool state.  So:
//   * per-lane output ring in LDS (dword-interleaved across lanes: conflict free); appends and ring reads
//     are byte-granular through v_perm_b32 on aligned dwords (no 64-bit shifts, no masking: bytes written
//     past the end of an append are overwritten by the next one); matches whose offset fits the ring are
//     served from LDS; finished output leaves 64 bytes at a time, four lanes storing one lane's line, so that a
//     store instruction covers 16 full lines instead of touching 64 (stage (c));
//   * every header byte (token, one literal-length byte, <= 11 literals, offset, one match-length byte)
//     comes out of a 32-byte register window over the compressed stream that slides 16 bytes at a time;
//     its loads are requested at least one header before they are needed;
//   * far matches and long literal runs stream through 16 (32 when more than 16 bytes are left) bytes of
//     registers that are requested at the END of an iteration and consumed in the next ones;
//   * the hot paths are straight-line code on integer state (selects, unconditional LDS accesses whose
//     result is discarded when not needed); only rare events are branches: length runs of 0xFF bytes,
//     the last bytes of the input, offset 0, errors, the final literal run.
#pragma once
#include "lz4hip_common.hpp"

namespace lz4hip {

constexpr int kChunkedRingBytes = 128;     // per-lane output ring
constexpr int kChunkedTableBytes = 128;    // byte-permute selectors of the periods 1..7, behind the 64 rings
constexpr int kChunkedFlushBytes = 1024;   // 64 flush records {lane, flushed, dst pointer}, behind the table
constexpr unsigned chunked_lds_bytes(int ring_bytes) { return 64u * (unsigned)ring_bytes + (unsigned)(kChunkedTableBytes + kChunkedFlushBytes); }

// what the next chunk of a lane's current copy is made from (>= kSlowLit: rare byte-wise sources)
enum ChunkMode { kIdle = 0, kReg = 1, kNear = 2, kGlobal = 3, kSlowLit = 4, kZeroOff = 5 };
// what has to be parsed / started once the current copy is finished
enum ChunkPending { kNeedToken = 0, kNeedMatch = 1, kNeedHeader = 2 };

// Entry p (1..7) of the period table: 16 selector bytes, byte b = b mod p (source byte of stream byte b).
// Written by lanes 0..31 of the wavefront (one dword each) before any lane decodes.
LZ4HIP_DEVICE void chunked_init_period_table(unsigned char* lds, int lane, int ring_bytes)
{
    if (lane < 32) {
        const uint32_t p = (uint32_t)lane >> 2, j = (uint32_t)lane & 3u;
        uint32_t v = 0;
        if (p) for (uint32_t b = 0; b < 4; b++) v |= ((4u * j + b) % p) << (8u * b);
        ((uint32_t*)(lds + 64 * ring_bytes))[lane] = v;
    }
    wv::mem_sync();
}

// All 64 lanes of the wavefront call this together and stay in the loop until the last one is done: a lane
// without a block (`active` false) or with a finished block still lends a hand to the cooperative flush.
template <bool KNOWN, int OUT_BYTES>
LZ4HIP_DEVICE int chunked_decode_block(unsigned char* lds, int lane, bool active, const uint8_t* __restrict__ src, int iend,
                                       uint8_t* dst, int oend)
{
    constexpr int RW = OUT_BYTES / 4;                                // ring dwords per lane
    static_assert(OUT_BYTES >= 128 && (OUT_BYTES & (OUT_BYTES - 1)) == 0, "ring: power of two, >= 128 bytes");
    // An append writes whole dwords, up to 19 bytes past its last byte; those land on ring bytes op-OUT_BYTES+19
    // and older.  Everything younger must stay intact: ring sources (<= kNearMax back) and unflushed output (< 91 back).
    constexpr int kNearMax = OUT_BYTES - 20;                         // largest offset served from the ring
    // dword k of this lane lives at LDS byte ((k & (RW-1)) << 8) | (lane << 2); RING_AT(p, j) is the dword j dwords
    // after the one containing output byte p (two instructions per address: v_lshl_add_u32, v_and_or_b32)
    const uint32_t lane4 = (uint32_t)lane << 2;
    constexpr uint32_t kRingMask = (uint32_t)(RW - 1) << 8;
    const uint32_t* period_tab = (const uint32_t*)(lds + 64 * OUT_BYTES);
    Aligned16* flush_rec = (Aligned16*)(lds + 64 * OUT_BYTES + kChunkedTableBytes);
#define RING_AT(p, j) (*(uint32_t*)(lds + (((((uint32_t)(p)) << 6) + 256u * (uint32_t)(j)) & kRingMask | lane4)))

    // ---- per-lane state (plain integers: bools would live in SGPR lane masks and cost s_and/s_or traffic) ----
    int ip = 0;                  // position of the next header to parse
    int op = 0, flushed = 0;     // bytes produced / bytes already stored to dst (multiple of 64)
    uint32_t tail = 0;           // ring dword containing op: its low (op & 3) bytes are output, the rest is junk
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;   // window src[win_pos .. win_pos+32) (valid iff win_ok)
    int win_pos = 0, win_ok = 0;
    int mode = kIdle, rem = 0;   // current copy: source kind and bytes left
    int stride = 16;             // bytes per chunk (16, a multiple of a short period, or the offset when source and chunk would overlap)
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;   // kReg: 16 bytes of the periodic stream of an offset < 8 match; kGlobal: fetched source bytes
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;   // kGlobal: the 16 bytes after (c0..c3) when a copy of > 16 bytes was requested
    int gcount = 0;              // kGlobal: fetched 16-byte pieces not yet consumed (0..2): c, then e
    const uint8_t* gptr = dst;   // kGlobal: where the next 16-byte fetch comes from
    int lit_src = 0;             // kSlowLit: position of the next literal byte in src
    int off = 8, ml = 0;         // pending / current match
    int pend = kNeedToken;
    uint32_t token = 0;
    int final_run = 0, result = 0;
    int done = 0;                // block finished (result holds the return value); the lane only helps flushing from now on
    if (!active || (!KNOWN && iend == 0)) { done = 1; final_run = 1; }   // lz4.c:946 returns -(0)

    // Append the low n_ bytes of the data dwords to the output ring at op.  The data is rotated to the byte
    // phase of op with one v_perm_b32 per dword and stored as whole dwords; only the first dword is merged
    // (with the bytes below op).  Bytes past n_ are junk that the next append overwrites.
#define APPEND_HEAD()                                                                                   \
        const uint32_t sb_ = (uint32_t)op & 3u;                                                         \
        /* byte b <- source byte 4 + b - sb: selector bytes (4-sb, 5-sb, 6-sb, 7-sb) */                 \
        const uint32_t s_ = wv::alignbyte(0x08070605u, 0x04030201u, sb_ ^ 3u);                          \
        const uint32_t keep_ = (1u << (8u * sb_)) - 1u;                                                 \
        const int k_ = op
#define APPEND_TAIL(n_)                                                                                 \
        op += (n_);                                                                                     \
        tail = RING_AT(op, 0)
#define APPEND4(d0_, d1_, d2_, d3_, n_)                                                                 \
    do {                                                                                                \
        APPEND_HEAD();                                                                                  \
        RING_AT(k_, 0) = (tail & keep_) | wv::perm(d0_, 0u, s_);                                              \
        RING_AT(k_, 1) = wv::perm(d1_, d0_, s_);                                                          \
        RING_AT(k_, 2) = wv::perm(d2_, d1_, s_);                                                          \
        RING_AT(k_, 3) = wv::perm(d3_, d2_, s_);                                                          \
        RING_AT(k_, 4) = wv::perm(0u, d3_, s_);                                                           \
        APPEND_TAIL(n_);                                                                                \
    } while (0)
#define APPEND3(d0_, d1_, d2_, n_)                                                                      \
    do {                                                                                                \
        APPEND_HEAD();                                                                                  \
        RING_AT(k_, 0) = (tail & keep_) | wv::perm(d0_, 0u, s_);                                              \
        RING_AT(k_, 1) = wv::perm(d1_, d0_, s_);                                                          \
        RING_AT(k_, 2) = wv::perm(d2_, d1_, s_);                                                          \
        RING_AT(k_, 3) = wv::perm(0u, d2_, s_);                                                           \
        APPEND_TAIL(n_);                                                                                \
    } while (0)
    // selector that extracts 4 bytes at byte phase (p & 3) from a dword pair: wv::perm(hi, lo, PHASE_SEL(p))
#define PHASE_SEL(p_) wv::alignbyte(0x07060504u, 0x03020100u, (uint32_t)(p_) & 3u)

    // Slide the window so that it covers [pos, pos + 16): usually nothing to do (a 16-byte load serves
    // ~3 short sequences); crossing into the second half shifts it and requests the next 16 bytes.
#define SLIDE_WINDOW(pos)                                                                               \
    do {                                                                                                \
        const int d_ = (pos) - win_pos;                                                                 \
        if (win_ok && d_ >= 0 && d_ < 16) {                                                             \
        } else if (win_ok && d_ >= 16 && d_ < 32 && win_pos + 48 <= iend) {                             \
            w0 = w4; w1 = w5; w2 = w6; w3 = w7; win_pos += 16;                                          \
            const Vec16 n_ = load_v16(src + win_pos + 16);                                              \
            w4 = n_.w[0]; w5 = n_.w[1]; w6 = n_.w[2]; w7 = n_.w[3];                                     \
        } else if ((pos) + 32 <= iend) {                                                                \
            win_pos = (pos); win_ok = 1;                                                                \
            const Vec16 m_ = load_v16(src + win_pos), n_ = load_v16(src + win_pos + 16);                \
            w0 = m_.w[0]; w1 = m_.w[1]; w2 = m_.w[2]; w3 = m_.w[3];                                     \
            w4 = n_.w[0]; w5 = n_.w[1]; w6 = n_.w[2]; w7 = n_.w[3];                                     \
        } else win_ok = 0;                                                                              \
    } while (0)

    if (!done) SLIDE_WINDOW(0);

    for (;;) {
        // =========================== (b) one chunk (<= 16 bytes) of the current copy ===========================
        {
            // ring source (read unconditionally; only used by kNear)
            const int sp = op - off;
            uint32_t r0 = RING_AT(sp, 0), r1 = RING_AT(sp, 1), r2 = RING_AT(sp, 2), r3 = RING_AT(sp, 3), r4 = RING_AT(sp, 4);
            const uint32_t sr = PHASE_SEL(sp);
            const bool can = rem > 0 && !(mode == kGlobal && gcount == 0);
            int n = can ? (rem < stride ? rem : stride) : 0;
            LZ4HIP_KEEP(r0); LZ4HIP_KEEP(r1); LZ4HIP_KEEP(r2); LZ4HIP_KEEP(r3); LZ4HIP_KEEP(r4);
            const bool near = mode == kNear;
            uint32_t v0 = near ? wv::perm(r1, r0, sr) : c0, v1 = near ? wv::perm(r2, r1, sr) : c1;
            uint32_t v2 = near ? wv::perm(r3, r2, sr) : c2, v3 = near ? wv::perm(r4, r3, sr) : c3;
            if (mode >= kSlowLit && can) {                           // rare byte-wise sources, 8 bytes at a time
                n = n < 8 ? n : 8;
                uint64_t acc = 0;
                if (mode == kSlowLit) { for (int b = 0; b < n; b++) if (lit_src + b < iend) acc |= (uint64_t)src[lit_src + b] << (8 * b); lit_src += n; }
                else                  { for (int b = 0; b < n; b++) acc |= (uint64_t)dst[op + b] << (8 * b); }   // offset 0: keep what dst holds
                v0 = (uint32_t)acc; v1 = (uint32_t)(acc >> 32);
            }
            const bool took = can && mode == kGlobal;                // c consumed: e (if there) becomes the next chunk
            c0 = took ? e0 : c0; c1 = took ? e1 : c1; c2 = took ? e2 : c2; c3 = took ? e3 : c3;
            gcount = took ? gcount - 1 : gcount;
            APPEND4(v0, v1, v2, v3, n);
            rem -= n;
            mode = rem == 0 ? (int)kIdle : mode;
        }

        // =========================== (a) header parsing ===========================
        int err = 0;                                                 // nonzero: this lane's stream is corrupt, value = return code
        if (rem == 0 && pend == kNeedHeader) {
            // ---- offset + match length at ip (after a literal run that did not fit the token's window) ----
            int p = ip + 2;
            bool have = false;
            if (win_ok) {
                const int d = ip - win_pos;
                const bool q2 = (d & 8) != 0, q1 = (d & 4) != 0;
                const uint32_t y0 = q2 ? w2 : w0, y1 = q2 ? w3 : w1, y2 = q2 ? w4 : w2;
                const uint32_t z0 = q1 ? y1 : y0, z1 = q1 ? y2 : y1;
                const uint32_t hx = wv::perm(z1, z0, PHASE_SEL(d));
                off = (int)(hx & 0xFFFFu);
                ml = (int)(token & 15u);
                const uint32_t b = (hx >> 16) & 255u;
                if (ml != 15) have = true;
                else if (b != 255u && (KNOWN || p < iend - (kLastLiterals + 1))) { ml += (int)b; p++; have = true; }
            }
            if (!have) {                                             // rare: byte-wise, lz4.c:862-866 / :979-997
                p = ip;
                off = (int)((p < iend ? src[p] : 0u) | ((p + 1 < iend ? src[p + 1] : 0u) << 8));
                p += 2;
                ml = (int)(token & 15u);
                if (ml == 15) {
                    if (KNOWN) {
                        uint32_t b;
                        while ((b = (p < iend ? src[p] : 0u)) == 255) { ml += 255; p++; if (ml > (1 << 30)) { err = -p; break; } }
                        ml += (int)b; p++;
                    } else {
                        while (p < iend - (kLastLiterals + 1)) { const uint32_t b = src[p]; p++; ml += (int)b; ml = ml > (1 << 30) ? (1 << 30) : ml; if (b != 255) break; }
                    }
                }
            }
            ml += kMinMatch;
            if (err != 0) {}
            else if (op - off < 0) err = -(ip + 2);
            else if ((int64_t)op + ml > (int64_t)oend - kLastLiterals) err = -p;
            ip = p;
            pend = kNeedMatch;
            SLIDE_WINDOW(ip);
        }
        if (rem == 0 && pend == kNeedToken && !final_run) {
            // ---- the 16 bytes at ip, from the window: dword shift network + one byte permute per dword ----
            const int d = ip - win_pos;
            const bool q2 = (d & 8) != 0, q1 = (d & 4) != 0;
            const uint32_t y0 = q2 ? w2 : w0, y1 = q2 ? w3 : w1, y2 = q2 ? w4 : w2, y3 = q2 ? w5 : w3, y4 = q2 ? w6 : w4, y5 = q2 ? w7 : w5;
            const uint32_t z0 = q1 ? y1 : y0, z1 = q1 ? y2 : y1, z2 = q1 ? y3 : y2, z3 = q1 ? y4 : y3, z4 = q1 ? y5 : y4;
            const uint32_t sx = PHASE_SEL(d);
            const uint32_t x0 = wv::perm(z1, z0, sx), x1 = wv::perm(z2, z1, sx), x2 = wv::perm(z3, z2, sx), x3 = wv::perm(z4, z3, sx);
            token = x0 & 255u;
            const uint32_t mlc = token & 15u;
            const uint32_t b1 = (x0 >> 8) & 255u;
            const bool ext1 = (token >> 4) == 15u;
            int ll = (int)(token >> 4) + (ext1 ? (int)b1 : 0);
            int pos = ip + 1 + (ext1 ? 1 : 0);                       // position after token (+ literal-length bytes)
            if (!win_ok || (ext1 && b1 == 255u)) {                   // rare: byte-wise, lz4.c:844 / :957-961
                token = ip < iend ? src[ip] : 0u;
                ll = (int)(token >> 4);
                pos = ip + 1;
                if (ll == 15) {
                    uint32_t b = 255;
                    if (KNOWN) { do { b = pos < iend ? src[pos] : 0u; pos++; ll += (int)b; if (ll > (1 << 30)) { err = -pos; ll = 0; break; } } while (b == 255); }
                    else       { while (pos < iend && b == 255) { b = src[pos]; pos++; ll += (int)b; ll = ll > (1 << 30) ? (1 << 30) : ll; } }   // saturate: the reference counts in size_t
                }
            }
            const bool in_win = win_ok && ll <= 11;                  // literals, offset and first match-length byte are in x0..x3
            const int lit_end = (int)((int64_t)op + ll > 0x7FFFFFFF ? 0x7FFFFFFF : op + ll);
            const bool last = KNOWN ? (lit_end > oend - 8) : (lit_end > oend - kMfLimit || pos + ll > iend - 8);
            const int lit_mode = in_win ? (int)kReg : (pos + ll + 16 <= iend ? (int)kGlobal : (int)kSlowLit);
            gptr = lit_mode == kGlobal ? src + pos : gptr;
            gcount = lit_mode == kGlobal ? 0 : gcount;
            lit_src = pos;
            stride = 16;
            // literals that sit in the window (bytes 1..11 of x) are appended right now; longer runs are streamed by stage (b)
            APPEND3(wv::alignbyte(x1, x0, 1), wv::alignbyte(x2, x1, 1), wv::alignbyte(x3, x2, 1), in_win ? ll : 0);
            rem = in_win ? 0 : ll;
            mode = rem ? lit_mode : (int)kIdle;
            if (last) {                                              // rare: final literal run, lz4.c:851-858 / :965-975
                if (KNOWN) { if (err == 0 && (lit_end != oend || pos + ll > iend)) err = -pos; }
                else       { if (lit_end > oend || pos + ll != iend) err = -pos; }
                final_run = 1;
                result = KNOWN ? pos + ll : lit_end;
            } else {
                if (KNOWN && err == 0 && pos + ll > iend) err = -pos;   // never read literals past the source
                // offset + match length from the same 16 bytes when they are all there
                const int e = 3 + ll;                                // index of the first match-length byte (<= 14 when in_win)
                const int o = 1 + ll, oq = o >> 2;                   // offset at bytes o, o+1 (o <= 12 when in_win), length byte at o+2
                const uint32_t xl = oq < 2 ? (oq == 0 ? x0 : x1) : (oq == 2 ? x2 : x3);
                const uint32_t xh = oq < 2 ? (oq == 0 ? x1 : x2) : x3;
                const uint32_t ot = wv::alignbyte(xh, xl, (uint32_t)o & 3u);   // bytes o .. o+3
                const uint32_t vo = ot & 0xFFFFu, extb = (ot >> 16) & 255u;
                // (the unknown-size decoder only reads a match-length byte while p < iend - 6, lz4.c:986)
                const bool fast = in_win && (mlc != 15u || (extb != 255u && (KNOWN || ip + e < iend - (kLastLiterals + 1))));
                const int p_off = ip + 3 + ll;                       // after the offset
                const int ml_fast = (int)mlc + kMinMatch + (mlc == 15u ? (int)extb : 0);
                const int ip_fast = p_off + (mlc == 15u ? 1 : 0);
                if (fast) {
                    off = (int)vo;
                    ml = ml_fast;
                    if (err == 0 && lit_end - off < 0) err = -p_off;               // lz4.c:863 / :980
                    if (err == 0 && lit_end + ml > oend - kLastLiterals) err = -ip_fast;   // lz4.c:893 / :1024
                }
                pend = fast ? (int)kNeedMatch : (int)kNeedHeader;
                ip = fast ? ip_fast : pos + ll;
                SLIDE_WINDOW(ip);                                    // a needed load travels while this sequence is copied
            }
        }
        if (err != 0) {                                              // corrupt stream: this lane is finished, nothing more is stored
            done = 1; final_run = 1; result = err;
            rem = 0; mode = kIdle; pend = kNeedToken;
        }
        if (rem == 0 && pend == kNeedMatch) {
            // ---- start the match copy: byte-wise semantics out[i] = out[i - off] ----
            // the 16-byte period of an offset < 8 match is built for every lane (straight-line) and kept if needed:
            // 8 ring bytes from op - off, then stream byte b = source byte (b mod off) via the selector table
            const bool periodic = off >= 1 && off < 8;
            const int osafe = periodic ? off : 1;
            const int sp = op - osafe;
            const uint32_t sr = PHASE_SEL(sp);
            const uint32_t r0 = RING_AT(sp, 0), r1 = RING_AT(sp, 1), r2 = RING_AT(sp, 2);
            const uint32_t s0 = wv::perm(r1, r0, sr), s1 = wv::perm(r2, r1, sr);
            const uint32_t* t = period_tab + 4 * osafe;
            const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
            c0 = periodic ? wv::perm(s1, s0, t0) : c0;
            c1 = periodic ? wv::perm(s1, s0, t1) : c1;
            c2 = periodic ? wv::perm(s1, s0, t2) : c2;
            c3 = periodic ? wv::perm(s1, s0, t3) : c3;
            const int l16 = 16 - (int)((0x24101000u >> (4 * osafe)) & 15u);   // off 1..7 -> 16,16,15,16,15,12,14
            // chunk size: a multiple of the period, or at most `off` when the source would overlap the chunk
            stride = periodic ? l16 : ((off >= 16 || off == 0) ? 16 : off);
            mode = off == 0 ? (int)kZeroOff : (periodic ? (int)kReg : (off <= kNearMax ? (int)kNear : (int)kGlobal));
            gptr = mode == kGlobal ? dst + (op - off) : gptr;        // older than the ring: already flushed
            gcount = mode == kGlobal ? 0 : gcount;
            rem = ml;
            pend = kNeedToken;
        }

        // =========================== (c) flush finished output, 64 bytes at a time, four lanes per line ===========================
        // (one lane storing its own 4 x 16 bytes makes every store instruction touch 64 different lines; measured
        //  ~12 CU-cycles per lane and store (tools/microbench_grouped.hip) -- the largest single cost of this kernel.
        //  Instead lanes with 64 finished bytes publish {lane, flushed, dst}; lanes 4m..4m+3 then store the line of
        //  the m-th publisher, 16 full 64-byte lines per store instruction.)
        {
            const bool need = !done && op - flushed >= 64;
            const uint64_t needy = wv::ballot(need);
            if (needy != 0) {                                        // wave-uniform
                const int cnt = wv::popc64(needy);
                if (need) {
                    const uint64_t dp = (uint64_t)dst;
                    flush_rec[wv::rank_below(needy)] = Aligned16{ { (uint32_t)lane, (uint32_t)flushed, (uint32_t)dp, (uint32_t)(dp >> 32) } };
                }
                wv::mem_sync();
                const int sub = lane & 3;
                for (int base = 0; base < cnt; base += 16) {         // wave-uniform trip count, almost always 1
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
                flushed += need ? 64 : 0;
            }
        }

        // =========================== (d) request the next 16 source bytes ===========================
        // (consumed in the next iteration; a far source lies > kNearMax behind op, the fetch reads 16 bytes
        //  from op - off, and at most 63 bytes behind op are still unflushed at this point)
        if (mode == kGlobal && gcount == 0 && rem > 0) {
            const Vec16 w = load_v16(gptr);
            c0 = w.w[0]; c1 = w.w[1]; c2 = w.w[2]; c3 = w.w[3];
            gcount = 1;
            if (rem > 16) {                                          // one memory round trip serves two chunks
                const Vec16 x = load_v16(gptr + 16);
                e0 = x.w[0]; e1 = x.w[1]; e2 = x.w[2]; e3 = x.w[3];
                gcount = 2;
            }
            gptr += 16 * gcount;
        }

        if (final_run && rem == 0 && !done) {
            // ---- end of block: write out the last (< 96) bytes exactly ----
            while (op - flushed >= 4) { const uint32_t q = RING_AT(flushed, 0); __builtin_memcpy(dst + flushed, &q, 4); flushed += 4; }
            if (flushed < op) {
                const uint32_t q = RING_AT(flushed, 0);
                for (int b = 0; flushed + b < op; b++) dst[flushed + b] = (uint8_t)(q >> (8 * b));
            }
            done = 1;
        }
        if (!wv::any(done == 0)) break;                              // every lane of the wavefront is finished
    }
    return result;
#undef RING_AT
#undef APPEND_HEAD
#undef APPEND_TAIL
#undef APPEND4
#undef APPEND3
#undef PHASE_SEL
#undef SLIDE_WINDOW
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.
// Static LDS: chunked_lds_bytes(OUT_BYTES) = 64 rings + the period table + the flush records.
template <bool KNOWN, int OUT_BYTES>
__global__ void __launch_bounds__(64) decode_chunked_kernel(Batch b, int filter)
{
    LZ4HIP_STATIC_LDS(lds, chunked_lds_bytes(OUT_BYTES));
    const int lane = (int)threadIdx.x;
    chunked_init_period_table(lds, lane, OUT_BYTES);
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
    const int r = chunked_decode_block<KNOWN, OUT_BYTES>(lds, lane, active, src, src_len, dst, out_size);
    if (active) b.result[blk] = r;
}

}  // namespace lz4hip
