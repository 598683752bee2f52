// lz4hip_decode.hpp -- batched LZ4 block decoders for gfx950, one wavefront per block.
//
// Replaces LZ4_uncompress (original/lz4.c:812-914 == LZ4_uncompress_64,
// src/LZ4pn/LZ4Codec.Unsafe64.Dirty.cs:534-661) and LZ4_uncompress_unknownOutputSize
// (original/lz4.c:916-1044 == src/LZ4pn/LZ4Codec.Unsafe64.Dirty.cs:667-806).
//
// Design (not a translation of the CPU copy loop):
//  * the sequence parse is wave-uniform and lives in SGPRs.  The compressed stream is held in a
//    512-byte register window (two dwords per lane, coalesced refills); token / length / offset
//    bytes are extracted with v_readlane, so the parse never waits on memory;
//  * a "short" sequence (literals + match <= 64 bytes) is materialised by ONE byte-per-lane store:
//    literal bytes come out of the register window via ds_bpermute, match bytes are gathered from
//    the already written output (global memory, L1/L2 resident) with the reference's byte-wise
//    overlap semantics out[i] = out[i - offset] expressed as a modulo on the lane index;
//  * long literal runs move 1 KiB per wave-instruction (16 B per lane);
//  * long matches use the periodicity of an overlapping LZ77 copy: once `offset` bytes exist, the
//    data is periodic, so the copy distance can be doubled every pass until it reaches 1 KiB and
//    the remainder streams at 16 B per lane even for offset 1 (e.g. an all-zero block).
//  Output is written straight to HBM.  A short sequence also leaves its bytes in a 4 KiB ring in LDS that mirrors the
//  wavefront's most recent output and takes its match bytes from there when the match starts inside it: on gfx9 one
//  counter (vmcnt) covers loads AND stores, so a gather from global memory waits for every store issued before it --
//  the write acknowledgement of the previous sequence, ~0.5 us -- while the ring's gather is an LDS access and the
//  stores to HBM are never waited for.  Matches that reach further back, and everything after a long copy until the
//  ring has filled up again, gather from global memory.  16 KiB of LDS per workgroup of four wavefronts leaves the
//  residency where the registers put it.
//
// Return values are the reference's: known-size decode returns the number of source bytes consumed
// or -(position of the error in the source); unknown-size decode returns bytes produced or
// -(position).  The only deliberate difference: the known-size decoder is also given the source
// length and never reads past it (the reference has undefined behaviour there); bytes past the end
// parse as zero and a literal run that would cross the end is reported as an error at its position.
#pragma once
#include "lz4hip_common.hpp"

#ifndef LZ4HIP_STAT
#define LZ4HIP_STAT(slot, cond) ((void)0)   /* the emulator build counts (tools/emu_decoder_stats.py) */
#endif
#ifndef LZ4HIP_STAT_ADD
#define LZ4HIP_STAT_ADD(slot, n) ((void)0)
#endif
// Section timers of the wavefront decoder (tools/dec_wave_sections.hip defines these to s_memtime accumulators; nothing in product builds).
#ifndef LZ4HIP_DEC_T0
#define LZ4HIP_DEC_DECL() ((void)0)
#define LZ4HIP_DEC_T0() ((void)0)
#define LZ4HIP_DEC_T(slot) ((void)0)
#define LZ4HIP_DEC_ADD(slot, n) ((void)0)
#define LZ4HIP_DEC_FLUSH() ((void)0)
#endif

namespace lz4hip {

// 512-byte sliding register window over the compressed stream of one block.
struct SrcWindow {
    const uint8_t* src;
    int len;        // bytes that may be read
    int base;       // window covers [base, base + 512), base % 256 == 0
    uint32_t w0, w1;

    LZ4HIP_DEVICE uint32_t fetch(int off) const
    {
        uint32_t v = 0;
        if (off + 4 <= len) {
            v = load_u32(src + off);
        } else {
            for (int k = 0; k < 4; k++)
                if (off + k < len) v |= (uint32_t)src[off + k] << (8 * k);
        }
        return v;
    }
    LZ4HIP_DEVICE void init(const uint8_t* s, int n)
    {
        src = s; len = n; base = 0;
        const int o = wv::lane() * 4;
        w0 = fetch(o);
        w1 = fetch(256 + o);
        wv::wait_vector_memory();
    }
    // make [p, p + n) addressable, n <= 256
    LZ4HIP_DEVICE void need(int p, int n)
    {
        if (p + n > base + 512) {
            const int nb = p & ~255;
            const int o = wv::lane() * 4;
            w0 = (nb == base + 256) ? w1 : fetch(nb + o);
            w1 = fetch(nb + 256 + o);
            base = nb;
            wv::wait_vector_memory();                                // (once per 256 bytes of input: see wait_vector_memory)
        }
    }
    // wave-uniform byte at wave-uniform position p
    LZ4HIP_DEVICE uint32_t peek(int p)
    {
        need(p, 1);
        const int idx = p - base, l = idx >> 2;
        const uint32_t a = wv::readlane(w0, l & 63), b = wv::readlane(w1, l & 63);
        return (((l & 64) ? b : a) >> ((idx & 3) * 8)) & 255u;
    }
    // the bytes at wave-uniform position p, p + 1, ... in the low bytes of a wave-uniform 64-bit value: at least FIVE of
    // them are valid (two window dwords, any byte phase); caller guarantees [p, p + 8) inside the window (need(p, 8))
    LZ4HIP_DEVICE uint64_t peek5(int p) const
    {
        const int idx = p - base, l = idx >> 2, l1 = l + 1;
        const uint32_t a = wv::readlane(w0, l & 63), b = wv::readlane(w1, l & 63);
        const uint32_t c = wv::readlane(w0, l1 & 63), d = wv::readlane(w1, l1 & 63);
        const uint64_t lo = (l & 64) ? b : a, hi = (l1 & 64) ? d : c;
        return ((hi << 32) | lo) >> ((idx & 3) * 8);
    }
    // per-lane byte at per-lane position p (caller guarantees p inside the window)
    LZ4HIP_DEVICE uint32_t gather(int p) const
    {
        const int idx = p - base, l = idx >> 2;
        const uint32_t a = wv::shuffle(w0, l & 63), b = wv::shuffle(w1, l & 63);
        return (((l & 64) ? b : a) >> ((idx & 3) * 8)) & 255u;
    }
};

// Overlapped LZ77 copy dst[pos + i] = dst[pos - off + i], i < n, off >= 1, by distance doubling.
// Every pass only reads bytes that earlier passes (or earlier sequences) have already written.
LZ4HIP_DEVICE void wave_match_copy(uint8_t* dst, int pos, int off, int n)
{
    const int lane = wv::lane();
    int dist = off;                    // always a multiple of off, <= bytes available behind `cur`
    int cur = pos, left = n;
    while (left > 0) {
        int step = left < dist ? left : dist;
        if (step > 1024) step = 1024;
        const uint8_t* from = dst + cur - dist;
        uint8_t* to = dst + cur;
        const int body = step & ~15;
        if (lane * 16 < body) store_v16(to + lane * 16, load_v16(from + lane * 16));
        const int t = body + lane;
        if (t < step) to[t] = from[t];
        wv::mem_sync();
        if (step == dist && dist < 1024) dist += dist;
        cur += step; left -= step;
        // A power-of-two period has reached exactly 1 KiB: every further KiB is the one just written, lane for lane.
        // Keep it in registers and only store (a run of zeros or of a short pattern becomes a fill, not a copy).
        if (dist == 1024 && left >= 1024) {
            const Vec16 v = load_v16(dst + cur - 1024 + lane * 16);
            for (; left >= 1024; left -= 1024, cur += 1024)              // (explicit 16-byte global stores: the compiler splits this one into dwords otherwise)
                wv::store_global16((uint64_t)(dst + cur + lane * 16), v.w[0], v.w[1], v.w[2], v.w[3]);
            wv::mem_sync();
        }
    }
}

struct alignas(8) BurstRec { uint32_t x, y; };   // a burst's sequence: token lane | literals << 8 | offset << 16; first output byte
constexpr int kWaveRingBytes = 4096;     // LDS mirror of a wavefront's most recent output (power of two)
constexpr int kWaveBurstRecBytes = 1024; // ... followed by the burst's sequence records (at most 64 x 8 bytes)
constexpr int kWaveLdsBytes = kWaveRingBytes + kWaveBurstRecBytes;

template <bool KNOWN>
LZ4HIP_DEVICE int decode_block(const uint8_t* src, int src_len, uint8_t* dst, int out_size, unsigned char* ring)
{
    const int lane = wv::lane();
    const int iend = src_len, oend = out_size;
    int ip = 0, op = 0;
    int ring_from = 0;                  // output bytes [max(ring_from, op - kWaveRingBytes), op) are in the ring

    if (!KNOWN && iend == 0) return 0;                      // original/lz4.c:946 returns -(0)
    SrcWindow win;
    win.init(src, src_len);

    // A BURST: a run of short sequences in one step (round 4).  The two paths below handle one sequence per trip of this loop,
    // ~100 wave-instructions and ~940 cycles of dependent latency each -- which is what a batch too small for the lane mapping
    // pays per sequence (a 64 KiB block of fuzzer-style data: 2.6 ms).  Here every lane parses the bytes at ip + lane AS IF a
    // sequence started there (token, <= 6 literals, offset, match length in the token: 78 % / 88 % of the sequences of such data
    // with <= 2 / <= 6 literals); a scalar walk from lane 0 follows the real sequence starts until 64 bytes of output are
    // accounted for; then lane j produces output byte j of the burst: a literal out of its sequence's token lane, or a match
    // byte from the LDS mirror of the recent output, or -- when its source lies inside the burst itself -- from the lane that
    // produces that byte, in as many rounds as the dependencies are deep (the reference's byte-wise overlap semantics
    // included: a match that overlaps itself is just a chain of such dependencies).  Anything the burst does not cover (length
    // bytes, long runs, sources behind the mirror, the end of the block, every error) leaves the burst to the paths below,
    // which remain the definition of the semantics.
    const int o_burst = oend - 96, i_burst = iend - 112;
    int burst_skip = 0, burst_fail = 0;                              // sequences to leave to the other paths after a burst that did not come about

    const int lane8 = (lane < 3 ? lane : 3) * 8;
    // While op <= o_safe and ip <= i_safe a sequence of at most 2 literals and a match of at most 18 bytes can neither be
    // the last one (lz4.c:851 / :965) nor run into the end of the output (:893 / :1024) nor read past the source.
    const int o_safe = KNOWN ? oend - 28 : oend - 32;
    const int i_safe = KNOWN ? iend - 5 : iend - 11;
    LZ4HIP_DEC_DECL();
    for (;;) {
        LZ4HIP_DEC_T0();
        if (burst_skip > 0) burst_skip--;
        else if (op <= o_burst && ip <= i_burst) {
            win.need(ip, 96);
            const int idx0 = ip - win.base, J0 = idx0 >> 2, ph = idx0 & 3;
            // this lane's seventeen bytes at ip + lane (token, <= 14 literals, offset; or token, <= 13 literals, offset, one length byte):
            // five window dwords, fetched from both halves of the window in ONE round of cross-lane reads.  (Until round 6: twelve bytes,
            // <= 6 literals and no length byte -- one sequence in eight of fuzzer-style data then left the burst for the general path,
            // 1 900 cycles each and a third of a block's time: profiles/r06/decoder_wave_sections_s_memtime.txt.)
            const int bi = ph + lane, dj = J0 + (bi >> 2);
            const uint32_t a0 = wv::shuffle(win.w0, dj & 63), b0 = wv::shuffle(win.w1, dj & 63);
            const uint32_t a1 = wv::shuffle(win.w0, (dj + 1) & 63), b1 = wv::shuffle(win.w1, (dj + 1) & 63);
            const uint32_t a2 = wv::shuffle(win.w0, (dj + 2) & 63), b2 = wv::shuffle(win.w1, (dj + 2) & 63);
            const uint32_t a3 = wv::shuffle(win.w0, (dj + 3) & 63), b3 = wv::shuffle(win.w1, (dj + 3) & 63);
            const uint32_t a4 = wv::shuffle(win.w0, (dj + 4) & 63), b4 = wv::shuffle(win.w1, (dj + 4) & 63);
            const uint32_t x0 = (dj & 64) ? b0 : a0, x1 = ((dj + 1) & 64) ? b1 : a1, x2 = ((dj + 2) & 64) ? b2 : a2,
                           x3 = ((dj + 3) & 64) ? b3 : a3, x4 = ((dj + 4) & 64) ? b4 : a4;
            const uint32_t sh = (uint32_t)bi & 3u;
            const uint32_t v0 = wv::alignbyte(x1, x0, sh), v1 = wv::alignbyte(x2, x1, sh), v2 = wv::alignbyte(x3, x2, sh),
                           v3 = wv::alignbyte(x4, x3, sh), v4 = x4 >> (8u * sh);          // bytes 0 .. 15, and byte 16 in the low byte of v4
            const uint32_t tok = v0 & 255u, ll = tok >> 4, mlc = tok & 15u;
            // offset (and the length byte behind it) at byte 1 + ll of that view
            const uint32_t fo = 1u + ll, fq = fo >> 2;                // (ll == 15: fq == 4, not simple, the values below are not used)
            const uint32_t flo = fq == 0u ? v0 : (fq == 1u ? v1 : (fq == 2u ? v2 : v3));
            const uint32_t fhi = fq == 0u ? v1 : (fq == 1u ? v2 : (fq == 2u ? v3 : v4));
            const uint32_t ot = wv::alignbyte(fhi, flo, fo & 3u);
            const uint32_t off = ot & 0xFFFFu, ext = (ot >> 16) & 255u;
            const bool longm = mlc == 15u;                            // match length 19 + one length byte (255 = more of them: general path)
            const bool simple = (ll <= 14u) & (off != 0u) & (!longm | ((ext != 255u) & (ll <= 13u)));
            const uint32_t olen_full = ll + mlc + (uint32_t)kMinMatch + (longm ? ext : 0u);
            const uint32_t olen = olen_full > 255u ? 255u : olen_full;            // (anything above 64 ends the walk)
            const uint32_t packed = olen | ((3u + ll + (longm ? 1u : 0u)) << 8);  // output bytes | input bytes of the sequence
            // the real sequence starts, from lane 0 on: a wave-uniform walk, one v_readlane per sequence
            const uint64_t simple_m = wv::ballot(simple);
            int s_tok = 0, tot = 0;
            uint64_t tok_m = 0, bound_m = 0;                         // token lanes of the burst; bit b: a sequence's output starts at byte b
            bool hit_other = false;                                  // the walk ended at a sequence the burst does not cover
            while (s_tok < 64) {
                if (!((simple_m >> s_tok) & 1ull)) { hit_other = true; break; }
                const uint32_t pk = wv::readlane(packed, s_tok);
                const int ol = (int)(pk & 255u);
                if (tot + ol > 64) break;
                tok_m |= 1ull << s_tok; bound_m |= 1ull << tot;
                tot += ol; s_tok += (int)(pk >> 8);
            }
            bool burst_done = false;
            if (tok_m != 0ull) {
                // token lanes leave a record per sequence in LDS (slot = its rank); output lane j reads the record of ITS sequence
                const bool is_tok = (tok_m >> lane) & 1ull;
                const uint32_t mine = is_tok ? olen : 0u;
                const uint32_t ost = wv::scan_add(mine) - mine;      // where this token's sequence starts in the burst's output
                BurstRec* const rec = (BurstRec*)(ring + kWaveRingBytes);
                if (is_tok) rec[wv::rank_below(tok_m)] = BurstRec{ (uint32_t)lane | (ll << 8) | (off << 16), ost };
                wv::mem_sync();
                const bool live = lane < tot;
                const int k = wv::rank_below(bound_m) + (int)((bound_m >> lane) & 1ull) - 1;   // the sequence of output byte `lane`
                const BurstRec r = rec[live ? k : 0];
                const int t_tl = (int)(r.x & 255u), t_ll = (int)((r.x >> 8) & 255u), t_off = (int)(r.x >> 16);
                const int rel = lane - (int)r.y;
                const bool is_lit = rel < t_ll;
                // a literal comes straight out of the register window: byte 1 + rel behind its sequence's token
                const uint32_t litb = win.gather(ip + t_tl + 1 + (is_lit ? rel : 0));
                const int srel = lane - t_off;                       // a match byte's source, relative to the burst's first byte
                const bool hist = live & !is_lit & (srel < 0);
                const int habs = op + srel;
                const int ring_lo = op - kWaveRingBytes > ring_from ? op - kWaveRingBytes : ring_from;
                if (!wv::any(hist & ((habs < ring_lo) | (habs < 0)))) {          // (a source before the block is an error: lz4.c:863 / :980, general path)
                    uint32_t val = litb;
                    bool res = is_lit | !live;
                    if (hist) { val = ring[habs & (kWaveRingBytes - 1)]; res = true; }
                    // sources inside the burst: pull from the lane that produces the byte, until nothing is left
                    int rounds = 0;
                    while (wv::any(!res)) {
                        rounds++;
                        const uint32_t sv = wv::shuffle(val, srel & 63), sr = wv::shuffle(res ? 1u : 0u, srel & 63);
                        if (!res && sr != 0u) { val = sv; res = true; }
                    }
#ifdef LZ4HIP_DEC_EXPERIMENT_NO_BURST_STORE                        /* (tools/dec_wave_sections.hip only: what do the burst's global store and the wait it causes cost?  wrong output) */
                    if (live) { ring[(op + lane) & (kWaveRingBytes - 1)] = (uint8_t)val; }
#else
                    if (live) { dst[op + lane] = (uint8_t)val; ring[(op + lane) & (kWaveRingBytes - 1)] = (uint8_t)val; }
#endif
                    wv::mem_sync();
                    LZ4HIP_STAT(20, lane == 0); LZ4HIP_STAT_ADD(21, lane == 0 ? wv::popc64(tok_m) : 0); LZ4HIP_STAT_ADD(22, lane == 0 ? rounds : 0); (void)rounds;
                    LZ4HIP_DEC_ADD(4, wv::popc64(tok_m)); LZ4HIP_DEC_ADD(5, rounds);
                    ip += s_tok; op += tot;
                    burst_done = true;
                    burst_fail = 0;
                    if (hit_other) burst_skip = 1;                   // the next sequence is known to need the general path
                }
            }
            if (burst_done) { LZ4HIP_DEC_T(0); continue; }
            LZ4HIP_DEC_T(1);
            LZ4HIP_STAT(23, lane == 0);
            burst_fail = burst_fail < 5 ? burst_fail + 1 : 5;       // no run of short sequences here: back off, 1, 2, 4 .. 32 sequences
            burst_skip = 1 << burst_fail >> 1;
            if (burst_skip < 1) burst_skip = 1;
        }
        // ---- the common short sequence in one step: at most two literals and a match whose length is in the token (4..18),
        //      whose source neither overlaps the match itself nor this sequence's literals.  Token, literals and offset are
        //      the five bytes at ip: one look at the register window, no length bytes, one byte-per-lane gather and store.
        //      Anything else -- longer runs, length bytes, overlapping matches, the last sequence, every error -- takes the
        //      general path below, which also remains the definition of the semantics. ----
        if (op <= o_safe && ip <= i_safe) {
            win.need(ip, 8);
            const uint64_t t5 = win.peek5(ip);
            const int tll = (int)((t5 >> 4) & 15u), tmlc = (int)(t5 & 15u);
            if (tll <= 2 && tmlc != 15) {
                const int t_lit_end = op + tll, t_ml = tmlc + kMinMatch;
                const int t_off = (int)((t5 >> (8 + 8 * tll)) & 0xFFFFu);
                if (t_off >= t_ml + tll && t_off <= t_lit_end) {
                    const int t_ref = t_lit_end - t_off;
                    const int ring_lo = op - kWaveRingBytes > ring_from ? op - kWaveRingBytes : ring_from;
                    const int sidx = t_ref + lane - tll;             // source of this lane's match byte (lanes >= tll)
                    const uint32_t lit = (uint32_t)(t5 >> 8) >> lane8;             // lanes < tll: their literal
                    // (two copies of the store on purpose: the one behind the LDS gather must not inherit the global
                    //  gather's vmcnt wait, which on gfx9 would also wait for the previous sequence's store)
                    if (t_ref >= ring_lo) {                                        // wave-uniform: the whole match is in the LDS mirror
                        uint32_t v = ring[sidx & (kWaveRingBytes - 1)];
                        if (lane < tll) v = lit;
                        if (lane < tll + t_ml) { dst[op + lane] = (uint8_t)v; ring[(op + lane) & (kWaveRingBytes - 1)] = (uint8_t)v; }
                    } else {
                        uint32_t v = lane >= tll && lane < tll + t_ml ? (uint32_t)dst[sidx] : 0u;
                        if (lane < tll) v = lit;
                        if (lane < tll + t_ml) { dst[op + lane] = (uint8_t)v; ring[(op + lane) & (kWaveRingBytes - 1)] = (uint8_t)v; }
                    }
                    wv::mem_sync();
                    LZ4HIP_STAT(24, lane == 0);
                    ip += 3 + tll; op = t_lit_end + t_ml;
                    LZ4HIP_DEC_T(2);
                    continue;
                }
            }
        }
        // ---- token + literal length: lz4.c:843-844 / :953-961 ----
        LZ4HIP_STAT(25, lane == 0);
        const uint32_t token = win.peek(ip); ip++;
        int ll = (int)(token >> 4);
        if (ll == 15) {
            uint32_t b = 255;
            if (KNOWN) { do { b = win.peek(ip); ip++; ll += (int)b; if (ll > (1 << 30)) return -ip; } while (b == 255); }
            else       { while (ip < iend && b == 255) { b = win.peek(ip); ip++; ll += (int)b; ll = ll > (1 << 30) ? (1 << 30) : ll; } }   // saturate: the reference counts in size_t
        }
        const int lit_end = (int)((int64_t)op + ll > 0x7FFFFFFF ? 0x7FFFFFFF : op + ll);

        // ---- last sequence (literals only): lz4.c:851-858 / :965-975 ----
        const bool last = KNOWN ? (lit_end > oend - 8) : (lit_end > oend - kMfLimit || ip + ll > iend - 8);
        if (last) {
            if (KNOWN) { if (lit_end != oend) return -ip; if (ip + ll > iend) return -ip; }
            else       { if (lit_end > oend) return -ip; if (ip + ll != iend) return -ip; }
            LZ4HIP_DEC_FLUSH();
            wave_copy(dst + op, src + ip, ll);
            return KNOWN ? ip + ll : lit_end;
        }
        if (KNOWN && ip + ll > iend) return -ip;           // never read literals past the source

        // ---- literal bytes of a short run, taken from the register window before it slides ----
        const bool short_lit = ll <= 64;
        uint32_t lit_byte = 0;
        if (short_lit && ll > 0) {
            win.need(ip, 66);
            lit_byte = win.gather(ip + (lane < ll ? lane : 0));
        }

        // ---- offset + match length: lz4.c:862-866 / :979-997 ----
        int p = ip + ll;
        const int off = (int)(win.peek(p) | (win.peek(p + 1) << 8));
        p += 2;
        const int ref = lit_end - off;
        if (ref < 0) return -p;
        int ml = (int)(token & 15);
        if (ml == 15) {
            if (KNOWN) {
                uint32_t b;
                while ((b = win.peek(p)) == 255) { ml += 255; p++; if (ml > (1 << 30)) return -p; }
                ml += (int)b; p++;
            } else {
                while (p < iend - (kLastLiterals + 1)) { const uint32_t b = win.peek(p); p++; ml += (int)b; ml = ml > (1 << 30) ? (1 << 30) : ml; if (b != 255) break; }
            }
        }
        ml += kMinMatch;
        if ((int64_t)lit_end + ml > (int64_t)oend - kLastLiterals) return -p;    // lz4.c:893 / :1024
        const int match_end = lit_end + ml;

        // ---- materialise the sequence ----
        // (offset 0 -- only in corrupt streams; the match bytes keep what dst holds, like the reference's copy from itself --
        //  takes the long path: it writes nothing to the ring and moves ring_from past the bytes the ring does not have)
        if (short_lit && ll + ml <= 64 && off != 0) {
            const int j = lane - ll;                         // index inside the match (valid when 0 <= j < ml)
            const bool in_match = j >= 0 && j < ml;
            int jj = j < 0 ? 0 : j;
            if (off < ml) jj = jj % off;                     // byte-wise overlap semantics
            const int sidx = ref + jj;                       // source index in dst
            const int from_lit = sidx - op;                  // >= 0: the byte is one of THIS sequence's literals
            const uint32_t via_lit = wv::shuffle(lit_byte, from_lit < 0 ? 0 : from_lit);
            const int ring_lo = op - kWaveRingBytes > ring_from ? op - kWaveRingBytes : ring_from;
            uint32_t v = lit_byte;
            if (in_match) v = from_lit >= 0 ? via_lit : (sidx >= ring_lo ? (uint32_t)ring[sidx & (kWaveRingBytes - 1)] : (uint32_t)dst[sidx]);
            if (lane < ll || in_match) { dst[op + lane] = (uint8_t)v; ring[(op + lane) & (kWaveRingBytes - 1)] = (uint8_t)v; }
        } else {
            ring_from = match_end;                           // the long copies below bypass the ring
            if (ll > 0) {
                if (short_lit) { if (lane < ll) dst[op + lane] = (uint8_t)lit_byte; }
                else wave_copy(dst + op, src + ip, ll);
                wv::mem_sync();
            }
            if (off != 0) wave_match_copy(dst, lit_end, off, ml);
        }
        wv::mem_sync();
        ip = p; op = match_end;
        LZ4HIP_DEC_T(3);
    }
}

// grid: ceil(n_blocks / waves_per_group) workgroups of 64 * waves_per_group threads.
constexpr int kWaveDecodeWavesPerGroup = 4;

template <bool KNOWN>
__global__ void __launch_bounds__(64 * kWaveDecodeWavesPerGroup) decode_kernel(Batch b, int filter)
{
    LZ4HIP_STATIC_LDS(rings, kWaveDecodeWavesPerGroup * kWaveLdsBytes);
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv::wave_in_block();
    if (blk >= b.n_blocks) return;
    const int src_len = wv::uniform(batch_src_len(b, blk));
    const int out_size = wv::uniform(batch_dst_cap(b, blk));
    if (!block_selected(filter, src_len, out_size)) return;
    const uint8_t* src = batch_src(b, blk);
    uint8_t* dst = batch_dst(b, blk);
    const int r = decode_block<KNOWN>(src, src_len, dst, out_size, rings + wv::wave_in_block() * kWaveLdsBytes);
    if (wv::lane() == 0) b.result[blk] = r;
}

}  // namespace lz4hip
