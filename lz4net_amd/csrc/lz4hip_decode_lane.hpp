// lz4hip_decode_lane.hpp -- batched LZ4 block decoders for gfx950, one LANE per block
// (64 independent blocks per wavefront).
//
// Same functions and return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).  Why a second mapping: decoding a
// block of short sequences (the reference's fuzzer data averages ~10 output bytes per sequence) is a
// chain of dependent memory round trips -- parse token, fetch the match source, store -- of roughly
// a microsecond each under load.  One wavefront per block exposes that latency 64 lanes wide for one
// sequence at a time; one lane per block keeps 64 sequences (of 64 blocks) in flight per wavefront
// and thousands per CU, which is what hides it.  The price is that every vector-memory instruction
// touches 64 different cache lines, so the copies are made as wide as the format allows:
//  * the token, up to 12 literal bytes and the 2-byte offset of a sequence arrive in ONE unaligned
//    16-byte load; the literals are stored straight out of that register (8- or 16-byte store);
//    the next sequence's 16-byte window is requested before the match copy starts, so the two
//    loads of a sequence overlap;
//  * the format guarantees 8 writable bytes after every non-final literal run and 5+ after every
//    match (lz4.c:851,887-893), so literal and match copies move 8/16 bytes at a time and may
//    over-write a few bytes that the same lane rewrites next;
//  * matches with offset < 8 are periodic: the period is built once in a 64-bit register and
//    stored with a stride that is a multiple of the offset -- no load per chunk.
// The last few bytes of a block (where an over-write could leave the block) use exact byte copies.
#pragma once
#include "lz4hip_common.hpp"

namespace lz4hip {

constexpr unsigned kLaneDecodeLdsBytes = 0;   // dynamic LDS reserved per 64-lane workgroup purely to bound residency (tuned on hardware)

LZ4HIP_DEVICE uint64_t load_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
LZ4HIP_DEVICE void store_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

template <bool KNOWN>
LZ4HIP_DEVICE int lane_decode_block(const uint8_t* __restrict__ src, int iend, uint8_t* dst, int oend)
{
    int ip = 0, op = 0;
    if (!KNOWN && iend == 0) return 0;                               // lz4.c:946 returns -(0)

    bool have_win = false;
    uint64_t lo = 0, hi = 0;
    for (;;) {
        // ---- 16-byte window at ip: token [+ literals + offset] ----
        if (!have_win && ip + 16 <= iend) {
            const Vec16 w = load_v16(src + ip);
            lo = w.w[0] | ((uint64_t)w.w[1] << 32);
            hi = w.w[2] | ((uint64_t)w.w[3] << 32);
            have_win = true;
        }
        const bool win = have_win;
        have_win = false;
        const uint32_t token = win ? (uint32_t)lo & 255u : (ip < iend ? src[ip] : 0u);
        ip++;
        int ll = (int)(token >> 4);
        const bool hdr = win && ll <= 12;                            // literals and offset are inside the window
        if (ll == 15) {                                              // lz4.c:844 / :957-961
            uint32_t b = 255;
            if (KNOWN) { do { b = ip < iend ? src[ip] : 0u; ip++; ll += (int)b; if (ll > (1 << 30)) return -ip; } while (b == 255); }
            else       { while (ip < iend && b == 255) { b = src[ip]; ip++; ll += (int)b; } }
        }
        const int lit_end = op + ll;

        // ---- final literal run: lz4.c:851-858 / :965-975 ----
        const bool last = KNOWN ? (lit_end > oend - 8) : (lit_end > oend - kMfLimit || ip + ll > iend - 8);
        if (last) {
            if (KNOWN) { if (lit_end != oend) return -ip; if (ip + ll > iend) return -ip; }
            else       { if (lit_end > oend) return -ip; if (ip + ll != iend) return -ip; }
            int k = 0;
            for (; k + 16 <= ll; k += 16) store_v16(dst + op + k, load_v16(src + ip + k));
            for (; k < ll; k++) dst[op + k] = src[ip + k];
            return KNOWN ? ip + ll : lit_end;
        }
        if (KNOWN && ip + ll > iend) return -ip;                     // never read literals past the source

        // ---- literals (lit_end <= oend - 8: an 8-byte over-write stays inside the block) ----
        if (hdr) {
            if (ll > 0) {
                store_u64(dst + op, (lo >> 8) | (hi << 56));
                if (ll > 8) store_u64(dst + op + 8, hi >> 8);
            }
        } else if (ll > 0) {
            if (ip + ll + 8 <= iend) {
                int k = 0;
                for (; k + 16 <= ll; k += 16) store_v16(dst + op + k, load_v16(src + ip + k));
                for (; k < ll; k += 8) store_u64(dst + op + k, load_u64(src + ip + k));
            } else {
                for (int k = 0; k < ll; k++) dst[op + k] = src[ip + k];
            }
        }

        // ---- offset + match length: lz4.c:862-866 / :979-997 ----
        int p = ip + ll;
        int off;
        if (hdr) {
            const int sh = 8 * (ll + 1);                             // 8 .. 104
            const uint64_t v = sh < 64 ? ((lo >> sh) | (hi << (64 - sh))) : (hi >> (sh - 64));
            off = (int)(v & 0xFFFFu);
        } else {
            off = (int)((p < iend ? src[p] : 0u) | ((p + 1 < iend ? src[p + 1] : 0u) << 8));
        }
        p += 2;
        const int ref = lit_end - off;
        if (ref < 0) return -p;
        int ml = (int)(token & 15);
        if (ml == 15) {
            if (KNOWN) {
                uint32_t b;
                while ((b = (p < iend ? src[p] : 0u)) == 255) { ml += 255; p++; if (ml > (1 << 30)) return -p; }
                ml += (int)b; p++;
            } else {
                while (p < iend - (kLastLiterals + 1)) { const uint32_t b = src[p]; p++; ml += (int)b; if (b != 255) break; }
            }
        }
        ml += kMinMatch;
        const int match_end = lit_end + ml;
        if (match_end > oend - kLastLiterals) return -p;             // lz4.c:893 / :1024

        // request the next sequence's window now: it travels while the match is copied
        if (p + 16 <= iend) {
            const Vec16 w = load_v16(src + p);
            lo = w.w[0] | ((uint64_t)w.w[1] << 32);
            hi = w.w[2] | ((uint64_t)w.w[3] << 32);
            have_win = true;
        }

        // ---- match: byte-wise semantics out[i] = out[i - off] ----
        uint8_t* to = dst + lit_end;
        const uint8_t* from = dst + ref;
        if (off == 0) {
            // out[i] = out[i]: the reference leaves whatever the buffer held
        } else if (match_end <= oend - 8) {
            if (off >= 8) {
                int k = 0;
                if (off >= 16) for (; k + 16 <= ml; k += 16) store_v16(to + k, load_v16(from + k));
                for (; k < ml; k += 8) store_u64(to + k, load_u64(from + k));
            } else {
                // periodic pattern of period `off` in a register; stride = largest multiple of off <= 8
                uint64_t pat = load_u64(from) & ((1ull << (8 * off)) - 1ull);
                int s = 8 * off;
                pat |= pat << s; s += s;
                if (s < 64) { pat |= pat << s; s += s; }
                if (s < 64) { pat |= pat << s; }
                const int stride = (int)((0x76586880u >> (4 * off)) & 15u);   // off 1..7 -> 8,8,6,8,5,6,7
                for (int k = 0; k < ml; k += stride) store_u64(to + k, pat);
            }
        } else {
            for (int k = 0; k < ml; k++) to[k] = from[k];            // tail of the block: exact
        }
        ip = p; op = match_end;
    }
}

// grid: ceil(n_blocks / 64) workgroups of 64 threads; lane i of workgroup g decodes block g*64 + i.
template <bool KNOWN>
__global__ void __launch_bounds__(64) decode_lane_kernel(Batch b, int filter)
{
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= b.n_blocks) return;
    const int src_len = batch_src_len(b, blk), out_size = batch_dst_cap(b, blk);
    if (!block_selected(filter, src_len, out_size)) return;
    b.result[blk] = lane_decode_block<KNOWN>(batch_src(b, blk), src_len, batch_dst(b, blk), out_size);
}

}  // namespace lz4hip
