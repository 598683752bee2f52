// lz4hip_decode_chunked.hpp -- lane-per-block LZ4 decoder as a CONVERGENT state machine:
// every loop iteration every lane (b) produces at most 16 output bytes from whatever source its state
// says, (a) parses a sequence header if its copy is finished -- appending the sequence's literals right
// away when they sit in the header window --, (c) flushes 64 bytes of finished output, (d) requests the
// next 16 source bytes if it is copying from global memory.
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
//   * per-lane output ring in LDS (qword-interleaved across lanes: conflict free); sequences are
//     appended exactly; matches whose offset fits the ring are served from LDS; finished output leaves
//     in 16-byte pieces that L2 merges into full lines;
//   * every header byte (token, one literal-length byte, <= 11 literals, offset, one match-length byte)
//     comes out of a 32-byte register window over the compressed stream that slides 16 bytes at a time;
//     its loads are requested at least one header before they are needed;
//   * far matches and long literal runs stream through a 16-byte register pair that is requested at
//     the END of an iteration and consumed in the next ones;
//   * the hot paths are straight-line code on integer state (selects, unconditional LDS accesses whose
//     result is discarded when not needed); only rare events are branches: length runs of 0xFF bytes,
//     the last bytes of the input, offset 0, errors, the final literal run.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_decode_lane.hpp"   // load_u64 / store_u64

namespace lz4hip {

constexpr int kChunkedRingBytes = 128;     // per-lane output ring (LDS = 64 x this per wavefront)

// what the next chunk of a lane's current copy is made from (>= kSlowLit: rare byte-wise sources)
enum ChunkMode { kIdle = 0, kReg = 1, kNear = 2, kGlobal = 3, kSlowLit = 4, kZeroOff = 5 };
// what has to be parsed / started once the current copy is finished
enum ChunkPending { kNeedToken = 0, kNeedMatch = 1, kNeedHeader = 2 };

template <bool KNOWN, int OUT_BYTES>
LZ4HIP_DEVICE int chunked_decode_block(unsigned char* lds, int lane, const uint8_t* __restrict__ src, int iend,
                                       uint8_t* dst, int oend)
{
    if (!KNOWN && iend == 0) return 0;                               // lz4.c:946 returns -(0)
    constexpr int OUT_Q = OUT_BYTES / 8;
    static_assert(OUT_BYTES >= 128, "ring too small for 16-byte appends + 64-byte flushes");
    constexpr int kNearMax = OUT_BYTES - 16;                         // largest offset served from the ring
    uint64_t* out_q = (uint64_t*)lds + lane;                         // qword k of this lane at out_q[(k & (OUT_Q-1)) * 64]
#define OUTQ(k) out_q[((k) & (OUT_Q - 1)) * 64]

    // ---- per-lane state (plain integers: bools would live in SGPR lane masks and cost s_and/s_or traffic) ----
    int ip = 0;                  // position of the next header to parse
    int op = 0, flushed = 0;     // bytes produced / bytes already stored to dst (multiple of 64)
    uint64_t tail = 0;           // qword containing op: low (op & 7) bytes valid, rest 0
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;   // 32-byte window over src[win_pos .. win_pos+32) (valid iff win_ok)
    int win_pos = 0, win_ok = 0;
    int mode = kIdle, rem = 0;   // current copy: source kind and bytes left
    int stride = 16;             // bytes per chunk (16, a multiple of a short period, or the offset when source and chunk would overlap)
    uint64_t cv = 0, cv2 = 0;    // kReg: 16 bytes of the periodic stream of an offset < 8 match
    uint64_t g0 = 0, g1 = 0;     // kGlobal: fetched source bytes
    int gcount = 0;              // kGlobal: valid 8-byte halves in (g0, g1)
    const uint8_t* gptr = dst;   // kGlobal: where the next 16-byte fetch comes from
    int lit_src = 0;             // kSlowLit: position of the next literal byte in src
    int off = 8, ml = 0;         // pending / current match
    int pend = kNeedToken;
    uint32_t token = 0;
    int final_run = 0, result = 0;

    // Append the low n (0..16) bytes of (vlo, vhi) to the output ring.  Up to three qwords are written; a qword
    // that receives nothing is redirected onto the current one, so no byte behind `op - 120` is ever touched.
#define APPEND(vlo_, vhi_, n_)                                                                          \
    do {                                                                                                \
        const int an_ = (n_);                                                                           \
        uint64_t al_ = (vlo_), ah_ = (vhi_);                                                            \
        al_ = an_ >= 8 ? al_ : (al_ & ((1ull << ((8 * an_) & 63)) - 1ull));                                    \
        ah_ = an_ >= 16 ? ah_ : (an_ > 8 ? (ah_ & ((1ull << (8 * (an_ & 7))) - 1ull)) : 0ull);          \
        const int ak_ = op >> 3, as_ = (op & 7) * 8, at_ = as_ + 8 * an_;                               \
        const uint64_t q0_ = tail | (al_ << as_);                                                       \
        const uint64_t q1_ = as_ ? (al_ >> ((64 - as_) & 63)) | (ah_ << as_) : ah_;                             \
        const uint64_t q2_ = as_ ? (ah_ >> ((64 - as_) & 63)) : 0ull;                                           \
        OUTQ(ak_) = q0_;                                                                                \
        OUTQ(at_ > 64 ? ak_ + 1 : ak_) = at_ > 64 ? q1_ : q0_;                                          \
        OUTQ(at_ > 128 ? ak_ + 2 : ak_) = at_ > 128 ? q2_ : q0_;                                        \
        tail = at_ < 64 ? q0_ : (at_ < 128 ? (at_ == 64 ? 0ull : q1_) : (at_ == 128 ? 0ull : q2_));    \
        op += an_;                                                                                      \
    } while (0)

    // Slide the window so that it covers [pos, pos + 16): usually nothing to do (a 16-byte load serves
    // ~3 short sequences); crossing into the second half shifts it and requests the next 16 bytes.
#define SLIDE_WINDOW(pos)                                                                               \
    do {                                                                                                \
        const int d_ = (pos) - win_pos;                                                                 \
        if (win_ok && d_ >= 0 && d_ < 16) {                                                             \
        } else if (win_ok && d_ >= 16 && d_ < 32 && win_pos + 48 <= iend) {                             \
            w0 = w2; w1 = w3; win_pos += 16;                                                            \
            const Vec16 n_ = load_v16(src + win_pos + 16);                                              \
            w2 = n_.w[0] | ((uint64_t)n_.w[1] << 32); w3 = n_.w[2] | ((uint64_t)n_.w[3] << 32);        \
        } else if ((pos) + 32 <= iend) {                                                                \
            win_pos = (pos); win_ok = 1;                                                                \
            const Vec16 m_ = load_v16(src + win_pos), n_ = load_v16(src + win_pos + 16);                \
            w0 = m_.w[0] | ((uint64_t)m_.w[1] << 32); w1 = m_.w[2] | ((uint64_t)m_.w[3] << 32);        \
            w2 = n_.w[0] | ((uint64_t)n_.w[1] << 32); w3 = n_.w[2] | ((uint64_t)n_.w[3] << 32);        \
        } else win_ok = 0;                                                                              \
    } while (0)

    SLIDE_WINDOW(0);

    for (;;) {
        // =========================== (b) one chunk (<= 16 bytes) of the current copy ===========================
        {
            const bool can = rem > 0 && !(mode == kGlobal && gcount == 0);
            int n = can ? (rem < stride ? rem : stride) : 0;
            // ring source (read unconditionally; only used by kNear)
            const int sp = op - off, ks = sp >> 3, ss = (sp & 7) * 8;
            const uint64_t r0 = OUTQ(ks), r1 = OUTQ(ks + 1), r2 = OUTQ(ks + 2);
            const uint64_t near_lo = ss ? (r0 >> ss) | (r1 << (64 - ss)) : r0;
            const uint64_t near_hi = ss ? (r1 >> ss) | (r2 << (64 - ss)) : r1;
            uint64_t vlo = mode == kReg ? cv : (mode == kGlobal ? g0 : near_lo);
            uint64_t vhi = mode == kReg ? cv2 : (mode == kGlobal ? g1 : near_hi);
            if (mode >= kSlowLit && can) {                           // rare byte-wise sources, 8 bytes at a time
                n = n < 8 ? n : 8;
                vlo = 0; vhi = 0;
                if (mode == kSlowLit) { for (int b = 0; b < n; b++) if (lit_src + b < iend) vlo |= (uint64_t)src[lit_src + b] << (8 * b); lit_src += n; }
                else                  { for (int b = 0; b < n; b++) vlo |= (uint64_t)dst[op + b] << (8 * b); }   // offset 0: keep what dst holds
            }
            gcount = (can && mode == kGlobal) ? 0 : gcount;
            APPEND(vlo, vhi, n);
            rem -= n;
            mode = rem == 0 ? (int)kIdle : mode;
        }

        // =========================== (a) header parsing ===========================
        int err = 0;                                                 // nonzero: this lane's stream is corrupt, value = return code
        if (rem == 0 && pend == kNeedToken && !final_run) {
            // ---- token [+ one literal-length byte] at ip, from the window ----
            const int d = ip - win_pos;
            const bool up = d >= 8;
            const uint64_t x0 = up ? w1 : w0, x1 = up ? w2 : w1, x2 = up ? w3 : w2;
            const int sw = 8 * (d & 7);
            const uint64_t lo = sw ? (x0 >> sw) | (x1 << (64 - sw)) : x0;
            const uint64_t hi = sw ? (x1 >> sw) | (x2 << (64 - sw)) : x1;
            token = (uint32_t)lo & 255u;
            const uint32_t mlc = token & 15u;
            const uint32_t b1 = (uint32_t)(lo >> 8) & 255u;
            const bool ext1 = (token >> 4) == 15u;
            int ll = (int)(token >> 4) + (ext1 ? (int)b1 : 0);
            int pos = ip + 1 + (ext1 ? 1 : 0);                       // position after token (+ literal-length bytes)
            if (!win_ok || (ext1 && b1 == 255u)) {                   // rare: byte-wise, lz4.c:844 / :957-961
                token = ip < iend ? src[ip] : 0u;
                ll = (int)(token >> 4);
                pos = ip + 1;
                if (ll == 15) {
                    uint32_t b = 255;
                    if (KNOWN) { do { b = pos < iend ? src[pos] : 0u; pos++; ll += (int)b; if (ll > (1 << 30)) return -pos; } while (b == 255); }
                    else       { while (pos < iend && b == 255) { b = src[pos]; pos++; ll += (int)b; } }
                }
            }
            const bool in_win = win_ok && ll <= 11;                  // literals, offset and first match-length byte are in (lo, hi)
            const int lit_end = op + ll;
            const bool last = KNOWN ? (lit_end > oend - 8) : (lit_end > oend - kMfLimit || pos + ll > iend - 8);
            const int lit_mode = in_win ? (int)kReg : (pos + ll + 16 <= iend ? (int)kGlobal : (int)kSlowLit);
            gptr = lit_mode == kGlobal ? src + pos : gptr;
            gcount = lit_mode == kGlobal ? 0 : gcount;
            lit_src = pos;
            stride = 16;
            // literals that sit in the window are appended right now; longer runs are streamed by stage (b)
            APPEND((lo >> 8) | (hi << 56), hi >> 8, in_win ? ll : 0);
            rem = in_win ? 0 : ll;
            mode = rem ? lit_mode : (int)kIdle;
            if (last) {                                              // rare: final literal run, lz4.c:851-858 / :965-975
                if (KNOWN) { if (lit_end != oend || pos + ll > iend) err = -pos; }
                else       { if (lit_end > oend || pos + ll != iend) err = -pos; }
                final_run = 1;
                result = KNOWN ? pos + ll : lit_end;
            } else {
                if (KNOWN && pos + ll > iend) err = -pos;            // never read literals past the source
                // offset + match length from the same 16 bytes when they are all there
                const int e = 3 + ll;                                // index of the first match-length byte (<= 14 when in_win)
                const uint32_t extb = (uint32_t)((e < 8 ? lo >> (8 * (e & 7)) : hi >> (8 * (e & 7))) & 255u);
                // (the unknown-size decoder only reads a match-length byte while p < iend - 6, lz4.c:986)
                const bool fast = in_win && (mlc != 15u || (extb != 255u && (KNOWN || ip + e < iend - (kLastLiterals + 1))));
                const int sh = 8 * ((1 + ll) & 15);                  // 8 .. 96 when in_win
                const uint64_t vo = sh < 64 ? ((lo >> sh) | (hi << ((64 - sh) & 63))) : (hi >> (sh & 63));
                const int p_off = ip + 3 + ll;                       // after the offset
                const int ml_fast = (int)mlc + kMinMatch + (mlc == 15u ? (int)extb : 0);
                const int ip_fast = p_off + (mlc == 15u ? 1 : 0);
                if (fast) {
                    off = (int)(vo & 0xFFFFu);
                    ml = ml_fast;
                    if (err == 0 && lit_end - off < 0) err = -p_off;               // lz4.c:863 / :980
                    if (err == 0 && lit_end + ml > oend - kLastLiterals) err = -ip_fast;   // lz4.c:893 / :1024
                }
                pend = fast ? (int)kNeedMatch : (int)kNeedHeader;
                ip = fast ? ip_fast : pos + ll;
                SLIDE_WINDOW(ip);                                    // a needed load travels while this sequence is copied
            }
        }
        if (rem == 0 && pend == kNeedHeader && err == 0) {
            // ---- offset + match length at ip (after a literal run that did not fit the token's window) ----
            int p = ip + 2;
            bool have = false;
            if (win_ok) {
                const int d = ip - win_pos;
                const bool up = d >= 8;
                const uint64_t x0 = up ? w1 : w0, x1 = up ? w2 : w1;
                const int sw = 8 * (d & 7);
                const uint64_t lo = sw ? (x0 >> sw) | (x1 << (64 - sw)) : x0;
                off = (int)((uint32_t)lo & 0xFFFFu);
                ml = (int)(token & 15u);
                const uint32_t b = (uint32_t)(lo >> 16) & 255u;
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
                        while ((b = (p < iend ? src[p] : 0u)) == 255) { ml += 255; p++; if (ml > (1 << 30)) return -p; }
                        ml += (int)b; p++;
                    } else {
                        while (p < iend - (kLastLiterals + 1)) { const uint32_t b = src[p]; p++; ml += (int)b; if (b != 255) break; }
                    }
                }
            }
            ml += kMinMatch;
            if (op - off < 0) err = -(ip + 2);
            else if (op + ml > oend - kLastLiterals) err = -p;
            ip = p;
            pend = kNeedMatch;
            SLIDE_WINDOW(ip);
        }
        if (err != 0) return err;
        if (rem == 0 && pend == kNeedMatch) {
            // ---- start the match copy: byte-wise semantics out[i] = out[i - off] ----
            // the 16-byte period of an offset < 8 match is built for every lane (straight-line) and kept if needed
            const int osafe = (off >= 1 && off < 8) ? off : 1;
            const int sp = op - osafe, ks = sp >> 3, ss = (sp & 7) * 8;
            const uint64_t q0 = OUTQ(ks), q1 = OUTQ(ks + 1);
            uint64_t pat = (ss ? (q0 >> ss) | (q1 << (64 - ss)) : q0) & ((1ull << (8 * osafe)) - 1ull);
            pat |= pat << (8 * osafe);                               // period x2 (<= 56-bit shift)
            pat |= osafe < 4 ? pat << (16 * osafe) : 0ull;           // x4 while it still fits
            pat |= osafe < 2 ? pat << 32 : 0ull;                     // x8 for offset 1
            // bytes 8..15 of the periodic stream: it also has period L8 = off * (8 / off) <= 8
            const int l8 = (int)((0x76586880u >> (4 * osafe)) & 15u);    // off 1..7 -> 8,8,6,8,5,6,7
            uint64_t pat_hi = l8 == 8 ? pat : (pat >> (8 * (8 - l8)));
            pat_hi |= l8 == 8 ? 0ull : (pat_hi << (8 * l8));
            const bool periodic = off >= 1 && off < 8;
            cv = periodic ? pat : cv; cv2 = periodic ? pat_hi : cv2;
            const int l16 = 16 - (int)((0x24101000u >> (4 * osafe)) & 15u);   // off 1..7 -> 16,16,15,16,15,12,14
            // chunk size: a multiple of the period, or at most `off` when the source would overlap the chunk
            stride = periodic ? l16 : ((off >= 16 || off == 0) ? 16 : off);
            mode = off == 0 ? (int)kZeroOff : (periodic ? (int)kReg : (off <= kNearMax ? (int)kNear : (int)kGlobal));
            gptr = mode == kGlobal ? dst + (op - off) : gptr;        // older than the ring: already flushed
            gcount = mode == kGlobal ? 0 : gcount;
            rem = ml;
            pend = kNeedToken;
        }

        // =========================== (c) flush finished output, 64 bytes at a time ===========================
        // (four back-to-back 16-byte stores fill whole 32/64-byte sectors: PMC showed 16-byte pieces issued
        //  iterations apart being written back to HBM as partial sectors, 2x the output bytes)
        if (op - flushed >= 64) {
            const int kq = flushed >> 3;
            for (int j = 0; j < 4; j++) {
                const uint64_t a = OUTQ(kq + 2 * j), b2 = OUTQ(kq + 2 * j + 1);
                const Vec16 v16 = { { (uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b2, (uint32_t)(b2 >> 32) } };
                store_v16(dst + flushed + 16 * j, v16);
            }
            flushed += 64;
        }

        // =========================== (d) request the next 16 source bytes ===========================
        // (consumed in the next iteration; a far source lies > OUT_BYTES - 16 behind op, the fetch reads 16 bytes
        //  from op - off, and at most 63 + 27 bytes behind op are still unflushed)
        if (mode == kGlobal && gcount == 0 && rem > 0) {
            const Vec16 w = load_v16(gptr);
            g0 = w.w[0] | ((uint64_t)w.w[1] << 32); g1 = w.w[2] | ((uint64_t)w.w[3] << 32);
            gcount = 2; gptr += 16;
        }

        if (final_run && rem == 0) {
            // ---- end of block: write out the last (< 96) bytes exactly ----
            while (op - flushed >= 8) { store_u64(dst + flushed, OUTQ(flushed >> 3)); flushed += 8; }
            if (flushed < op) {
                const uint64_t q = OUTQ(flushed >> 3);
                for (int b = 0; flushed + b < op; b++) dst[flushed + b] = (uint8_t)(q >> (8 * b));
            }
            return result;
        }
    }
#undef OUTQ
#undef APPEND
#undef SLIDE_WINDOW
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.  Dynamic LDS: 64 * OUT_BYTES.
template <bool KNOWN, int OUT_BYTES>
__global__ void __launch_bounds__(64) decode_chunked_kernel(Batch b, int filter)
{
    LZ4HIP_DYN_LDS(lds);
    const int lane = (int)threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x * 64 + lane;
    if (blk >= b.n_blocks) return;
    const int src_len = batch_src_len(b, blk), out_size = batch_dst_cap(b, blk);
    if (!block_selected(filter, src_len, out_size)) return;
    b.result[blk] = chunked_decode_block<KNOWN, OUT_BYTES>(lds, lane, batch_src(b, blk), src_len, batch_dst(b, blk), out_size);
}

}  // namespace lz4hip
