// lz4hip_decode_staged.hpp -- lane-per-block LZ4 decoder with a per-lane OUTPUT ring in LDS.
//
// Same functions / return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).
//
// Why: with one lane per block (lz4hip_decode_lane.hpp) the arithmetic is free but every 8..16-byte
// access is its own L2 request, and with half a million streams in flight the 128-byte lines those
// requests fall into do not survive in L2: rocprof showed ~14x the algorithmic bytes crossing the
// fabric, most of it partial-line output writes and match-source re-fetches.  Here each lane owns a
// small ring in LDS holding its most recent output:
//   * sequences are APPENDED to the ring (exact bytes, no over-writing copies);
//   * a completed 128-byte line is flushed with eight back-to-back 16-byte stores, which L2 merges into
//     one full-line write;
//   * a match whose offset fits the ring (77 % / 87 % / 95 % of the matches of fuzzer-style data for
//     256 / 512 / 1024-byte rings) is served from LDS and never touches global memory; older sources are
//     read back from the already flushed output.
// The ring is qword-interleaved across the 64 lanes (qword k of lane l at (k*64 + l)*8), so arbitrary
// per-lane positions are bank-conflict free for ds_read/write_b64.  The compressed stream is read as
// in the plain lane decoder: one unaligned 16-byte window per sequence, requested one sequence ahead.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_decode_lane.hpp"   // load_u64 / store_u64

namespace lz4hip {

constexpr bool kStagedByDefault = false;   // which lane-per-block decoder launch_decode() picks (tuned on hardware)
constexpr int kStagedRingBytes = 512;       // per-lane output ring (LDS use = 64 x this per wavefront)

template <int OUT_BYTES>
struct LaneStage {
    uint64_t* out_q;                 // LDS, already offset by the lane: qword k at out_q[(k & (OUT_Q-1)) * 64]
    static constexpr int OUT_Q = OUT_BYTES / 8;
    uint8_t* dst;
    int op;                          // bytes produced
    int flushed;                     // bytes [0, flushed) are in global memory (multiple of 128)
    uint64_t tail;                   // qword containing position op: low (op & 7) bytes valid, rest 0

    LZ4HIP_DEVICE uint64_t out_qword(int k) const { return out_q[(k & (OUT_Q - 1)) * 64]; }

    LZ4HIP_DEVICE void init(unsigned char* lds, int lane, uint8_t* d)
    {
        out_q = (uint64_t*)lds + lane;
        dst = d; op = 0; flushed = 0; tail = 0;
    }
    // 8 recent output bytes at position p (op - OUT_BYTES + 16 <= p; bytes at or past op are garbage)
    LZ4HIP_DEVICE uint64_t out8(int p) const
    {
        const int k = p >> 3, s = (p & 7) * 8;
        const uint64_t q0 = out_qword(k), q1 = out_qword(k + 1);
        return s ? (q0 >> s) | (q1 << (64 - s)) : q0;
    }

    // append the low n (1..8) bytes of v to the output
    LZ4HIP_DEVICE void append(uint64_t v, int n)
    {
        if (n < 8) v &= (1ull << (8 * n)) - 1ull;
        const int k = op >> 3, s = (op & 7) * 8;
        const uint64_t cur = tail | (v << s);
        out_q[(k & (OUT_Q - 1)) * 64] = cur;
        if (s + 8 * n >= 64) {
            tail = s ? (v >> (64 - s)) : 0ull;
            if ((s + 8 * n) > 64) out_q[((k + 1) & (OUT_Q - 1)) * 64] = tail;
        } else {
            tail = cur;
        }
        op += n;
        if (op - flushed >= 128) flush_line();
    }
    // append n bytes read from p (a literal run of the compressed stream); `room` = bytes readable at p
    LZ4HIP_DEVICE void append_from(const uint8_t* __restrict__ p, int n, int room)
    {
        int k = 0;
        for (; k + 8 <= n && k + 8 <= room; k += 8) append(load_u64(p + k), 8);
        if (k < n) {
            // tail (and anything that may not be read 8 bytes wide): byte loads, still 8-byte appends
            for (; k < n; k += 8) {
                const int m = n - k < 8 ? n - k : 8;
                uint64_t v = 0;
                for (int b = 0; b < m; b++) if (k + b < room) v |= (uint64_t)p[k + b] << (8 * b);
                append(v, m);
            }
        }
    }
    LZ4HIP_DEVICE void flush_line()
    {
        const int k0 = flushed >> 3;
        for (int j = 0; j < 8; j++) {
            const uint64_t a = out_qword(k0 + 2 * j), b = out_qword(k0 + 2 * j + 1);
            Vec16 v = { { (uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32) } };
            store_v16(dst + flushed + 16 * j, v);
        }
        flushed += 128;
    }
    // write out whatever is left (< 128 bytes), exactly
    LZ4HIP_DEVICE void flush_rest()
    {
        int p = flushed;
        for (; p + 8 <= op; p += 8) store_u64(dst + p, out_qword(p >> 3));
        if (p < op) {
            const uint64_t q = out_qword(p >> 3);
            for (int b = 0; p + b < op; b++) dst[p + b] = (uint8_t)(q >> (8 * b));
        }
        flushed = op;
    }
};

template <bool KNOWN, int OUT_BYTES>
LZ4HIP_DEVICE int staged_decode_block(unsigned char* lds, int lane, const uint8_t* __restrict__ src, int iend,
                                      uint8_t* dst, int oend)
{
    if (!KNOWN && iend == 0) return 0;                               // lz4.c:946 returns -(0)
    LaneStage<OUT_BYTES> st;
    st.init(lds, lane, dst);
    constexpr int kNear = OUT_BYTES - 16;                            // largest offset served from the ring
    int ip = 0;
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
        const int lit_end = st.op + ll;

        // ---- final literal run: lz4.c:851-858 / :965-975 ----
        const bool last = KNOWN ? (lit_end > oend - 8) : (lit_end > oend - kMfLimit || ip + ll > iend - 8);
        if (last) {
            if (KNOWN) { if (lit_end != oend) return -ip; if (ip + ll > iend) return -ip; }
            else       { if (lit_end > oend) return -ip; if (ip + ll != iend) return -ip; }
            st.append_from(src + ip, ll, iend - ip);
            st.flush_rest();
            return KNOWN ? ip + ll : lit_end;
        }
        if (KNOWN && ip + ll > iend) return -ip;                     // never read literals past the source

        // ---- literals ----
        if (hdr) {
            if (ll > 0) {
                st.append((lo >> 8) | (hi << 56), ll < 8 ? ll : 8);
                if (ll > 8) st.append(hi >> 8, ll - 8);
            }
        } else {
            st.append_from(src + ip, ll, iend - ip);
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
        if (lit_end - off < 0) return -p;
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
        if (lit_end + ml > oend - kLastLiterals) return -p;          // lz4.c:893 / :1024

        // request the next sequence's window now: it travels while the match is copied
        if (p + 16 <= iend) {
            const Vec16 w = load_v16(src + p);
            lo = w.w[0] | ((uint64_t)w.w[1] << 32);
            hi = w.w[2] | ((uint64_t)w.w[3] << 32);
            have_win = true;
        }

        // ---- match: byte-wise semantics out[i] = out[i - off] ----
        if (off == 0) {
            // out[i] = out[i]: the reference leaves whatever the buffer held.  The staged output must still
            // advance; the bytes written are whatever the destination already contains.
            for (int k = 0; k < ml; k += 8) {
                const int n = ml - k < 8 ? ml - k : 8;
                uint64_t v = 0;
                for (int b = 0; b < n; b++) v |= (uint64_t)dst[st.op + b] << (8 * b);
                st.append(v, n);
            }
        } else if (off < 8) {
            // periodic: build the period once, append a multiple of `off` bytes per step
            uint64_t pat = st.out8(st.op - off) & ((1ull << (8 * off)) - 1ull);
            int s = 8 * off;
            pat |= pat << s; s += s;
            if (s < 64) { pat |= pat << s; s += s; }
            if (s < 64) { pat |= pat << s; }
            const int stride = (int)((0x76586880u >> (4 * off)) & 15u);     // off 1..7 -> 8,8,6,8,5,6,7
            for (int k = 0; k < ml; k += stride) st.append(pat, ml - k < stride ? ml - k : stride);
        } else if (off <= kNear) {
            for (int k = 0; k < ml; k += 8) st.append(st.out8(st.op - off), ml - k < 8 ? ml - k : 8);
        } else {
            // source is older than the ring: it has been flushed (flushed >= op - 127 > source end)
            const uint8_t* from = dst + (st.op - off);
            int k = 0;
            for (; k + 16 <= ml; k += 16) {
                const Vec16 v = load_v16(from + k);
                st.append(v.w[0] | ((uint64_t)v.w[1] << 32), 8);
                st.append(v.w[2] | ((uint64_t)v.w[3] << 32), 8);
            }
            for (; k < ml; k += 8) st.append(load_u64(from + k), ml - k < 8 ? ml - k : 8);
        }
        ip = p;
    }
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.
// Dynamic LDS: 64 * OUT_BYTES bytes.
template <bool KNOWN, int OUT_BYTES>
__global__ void __launch_bounds__(64) decode_staged_kernel(Batch b, int filter)
{
    LZ4HIP_DYN_LDS(lds);
    const int lane = (int)threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x * 64 + lane;
    if (blk >= b.n_blocks) return;
    const int src_len = batch_src_len(b, blk), out_size = batch_dst_cap(b, blk);
    if (!block_selected(filter, src_len, out_size)) return;
    b.result[blk] = staged_decode_block<KNOWN, OUT_BYTES>(lds, lane, batch_src(b, blk), src_len, batch_dst(b, blk), out_size);
}

}  // namespace lz4hip
