// lz4hip_decode_lane4.hpp -- lane-per-block LZ4 decoder, fourth generation.
//
// Round 4 measured what bounds the third generation (lz4hip_decode_lane3.hpp) with a MEMORY SKELETON of it
// (tools/decode_skeleton.hip: the kernel's global accesses for fuzzer-style / record-style data, no parsing):
// the skeleton runs at 830 GB/s (D2) / 696 GB/s (D3) -- the real kernel's 829 / 698.  Generation 3 sits on the ceiling the
// memory system sets for ITS ACCESS PATTERN, whatever its arithmetic costs; the same skeleton says what moves the ceiling
// (profiles/r04/decode_skeleton_*.txt):
//     input fetched as whole 64-byte sectors, once (not as two 32-byte halves 14 us apart)        +15 %
//     192 bytes of output ring per lane instead of 128 (a quarter of the far-match fetches gone)   +15 %
//     finished output leaving in 128-byte units (both sectors of a line in one store instruction)  +4 % (ring 192) .. +13 % (256)
// All three need LDS that generation 3 spends on its 64-byte-per-lane input staging ring.  So here the INPUT WINDOW LIVES IN
// REGISTERS (the kernel runs at three wavefronts per SIMD = 168 VGPRs, generation 3 used 104):
//   * W[4 + P/4]: the last 16 bytes of the previous piece and one whole piece (P = 64: a sector) of the compressed
//     stream, L[P/4]: the next piece, loaded by the lane itself with P/16 always-issued, predicated 16-byte loads as soon as
//     the previous L has moved into W.  No staging ring, no load records, no helper lanes, no landing pass.
//   * the 16 bytes at the input cursor (byte offset d into W) come out of W through a binary tree of v_cndmask_b32
//     (log2(P/4) levels; measured: 4.5 SIMD-cycles per select, tools/microbench_select_tree.hip) and one byte rotation;
//     the offset field of a sequence is picked out of those 16 bytes the same way.
//   * LDS per wavefront = 64 x R + 512 bytes: R = 192 at the twelve wavefronts per CU generation 3 had with R = 128.
// Round 5 (profiles/r05/decoder_sector_input_and_dual_stores_ab.txt; D2 at 2^20 blocks 1020 -> 1040-1064 GB/s, D3 825 -> 833-840):
//   * SECTOR INPUT (POL bit 4, the default): L is a whole aligned 64-byte sector -- four loads, ONE request to the memory side -- and
//     feeds W a 32-byte half at a time.  Window and select tree stay those of 32-byte pieces (+8 selects for the choice of the half), but a
//     sector of the source is fetched once instead of as two halves ~10 us apart, the second after its line had left the L2: 0.5 G of a
//     launch's 3.7 G requests gone.
//   * RING ROWS STORED TWICE instead of wrapped (LZ4HIP_DEC4_DUAL_STORE): gfx950 drops a DS store outside the workgroup's allocation
//     (tools/lds_out_of_range.hip), so the rows of an append go to `row` and to `row - ring size` (ds_write2st64_b32: two rows per
//     instruction) and the three instructions per row address that a ring of 48 rows cost are gone: -19 of ~305 vector-ALU instructions per
//     iteration.  Worth nothing while the request ceiling held (round 4), +2.5 % behind the sector input.
// Everything else is generation 3's design: dword-interleaved output ring, appends by DS_MSKOR_B32 + v_perm_b32, cooperative
// flush (four lanes per 64-byte line, or eight per 128-byte unit), far matches fetched from the lane's own output one and a
// half iterations ahead, parse-ahead of one sequence, a byte-wise parser behind one wave-level branch, hand-counted vmcnt.
//
// Same functions / return conventions as lz4hip_decode.hpp (LZ4_uncompress, original/lz4.c:812-914;
// LZ4_uncompress_unknownOutputSize, original/lz4.c:916-1044).
#pragma once
#include "lz4hip_common.hpp"
#include <type_traits>

#ifndef LZ4HIP_ITERATION_HOOK
#define LZ4HIP_ITERATION_HOOK(lane) ((void)0)
#endif
#ifndef LZ4HIP_STAT
#define LZ4HIP_STAT(slot, cond) ((void)0)   /* the emulator build counts lane-iterations per state (tools/emu_decoder_stats.py) */
#endif
// Section markers for the per-section instruction table of the compiled kernel (tools/isa_lane4_table.py builds with
// -DLZ4HIP_SECTION_MARKERS: each marker becomes a comment line in the assembly listing; volatile asm statements -- most of this kernel's
// selects, loads and stores -- keep their order relative to the markers).  Nothing in product builds.
#ifdef LZ4HIP_SECTION_MARKERS
#define LZ4HIP_SECTION(name) asm volatile("; @@SECTION " name)
#else
#define LZ4HIP_SECTION(name) ((void)0)
#endif

namespace lz4hip {

#ifndef LZ4HIP_DEC4_FLUSH_RECS
#define LZ4HIP_DEC4_FLUSH_RECS 0      /* 0 = all the two store instructions can carry; the emulator's 'starved' build: 4 */
#endif
#ifndef LZ4HIP_DEC4_DUAL_STORE
#define LZ4HIP_DEC4_DUAL_STORE 2      /* 1, 2: a ring store is issued at `row` and at `row - ring size`, the hardware drops the one outside the allocation (2: two rows per LDS instruction); 0: the row is wrapped */
#endif
constexpr unsigned lane4_lds_bytes(int ring_bytes) { return 64u * (unsigned)ring_bytes + 16u * 32u; }

// What a lane does when its block is finished.  NoNext: nothing (one block per lane, the wavefront ends when its last lane does).
// A persistent kernel passes a functor that stores the finished block's result and hands the lane its next block.
struct NoNext {
    static constexpr bool kPersistent = false;
    LZ4HIP_DEVICE bool operator()(int, const uint8_t*&, int&, uint8_t*&, int&) const { return false; }
};

enum L4Kind { kK4None = 0, kK4Near = 1, kK4Far = 2, kK4Lit = 3, kK4Zero = 4 };
enum L4Flag { kF4Final = 1, kF4Err = 2, kF4Header = 4 };   // pending sequence: final literal run / corrupt stream / no match yet (its header follows the literals)

// All 64 lanes of the wavefront call this together and stay in the loop until the last one is done.
//   R      bytes of output ring per lane (multiple of 16)
//   P      bytes of input per piece (32 or 64), loaded by the lane itself
//   FU     flush unit: 64 (one line, four lanes) or 128 (two adjacent lines, eight lanes)
//   FS     flush store instructions per flushing iteration (1 or 2): each carries 64 / (FU / 16) units
//   FE     the flush runs in every FE-th iteration (1 or 2)
//   IE     2: the next input piece is only requested in the iterations that do not flush (needs FE == 2); 1: in every iteration
//   POL    cache policy of the loads (wv::vm_load16_pred): low two bits = far-match fetches, next two bits = input pieces;
//          bit 4 (16): SECTOR INPUT -- L holds a whole aligned 64-byte sector (four loads, one request to the memory side) and feeds
//          W one 32-byte half at a time (needs P == 32): the window and its select tree stay those of 32-byte pieces, but each
//          sector of the source is fetched once instead of as two halves ~10 us apart (by then the first one's line has left the L2);
//          bit 5 (32): ring rows are WRAPPED instead of stored twice -- no DS store ever leaves the allocation.  The library launches this
//          instantiation on a device whose load-time probe (lds_drop_probe_kernel below) did not confirm that out-of-range stores are dropped
//   WAVES  wavefronts of the workgroup (1 or 4) that share `lds`: the rings of ALL of them are dword-interleaved -- row r of lane l of wavefront w at
//          r * 256 * WAVES + (64 w + l) * 4 -- so that the ring still starts at LDS address 0 and a row stored at `row - ring size`, or past the last row,
//          still falls outside the allocation (or into this wavefront's own flush records, see frec) whichever wavefront stores it; `wave` = w
template <bool KNOWN, int R, int P, int FU, int FS, int FE = 1, int IE = 1, int POL = 0, class NEXT = NoNext, int WAVES = 1>
LZ4HIP_DEVICE int lane4_decode_block(unsigned char* lds, int lane, bool active, const uint8_t* src, int iend,
                                     uint8_t* dst, int oend, NEXT next = NEXT(), int wave = 0)
{
    static_assert(WAVES == 1 || WAVES == 4, "one or four wavefronts per workgroup");
    constexpr uint32_t kRow = 256u * (uint32_t)WAVES;                // bytes from one ring row to the next
    static_assert(R >= 128 && R % 16 == 0 && R <= 1008, "ring: a multiple of 16 bytes, 128 .. 1008");
    static_assert(P == 32 || P == 64, "input piece: 32 or 64 bytes");
    static_assert(FU == 64 || (FU == 128 && R >= 192), "flush unit: 64 bytes, or 128 with a ring of at least 192");
    static_assert(FS == 1 || FS == 2, "one or two flush store instructions per iteration");
    static_assert(FE == 1 || FE == 2, "the flush runs in every iteration or in every second one");
    static_assert(IE == 1 || (IE == 2 && FE == 2), "input requests in every iteration, or alternating with the flush");
    constexpr int RW = R / 4;                                        // ring rows (one dword per lane per row)
    constexpr bool RPOW2 = (RW & (RW - 1)) == 0;
    constexpr uint32_t kRingBytes = (uint32_t)RW * kRow;             // the 64 * WAVES rings, dword-interleaved: row r of lane l at r * kRow + (64 * wave + l) * 4
    constexpr uint32_t kLdsBytes = lane4_lds_bytes(R) * (uint32_t)WAVES;
    constexpr bool LS = (POL & 16) != 0;                             // L = one aligned 64-byte sector, consumed as two pieces
    constexpr int kDual = (POL & 32) ? 0 : (LZ4HIP_DEC4_DUAL_STORE); // POL bit 5: ring rows WRAPPED, no store outside the allocation (the fallback of lz4hip_api.hip)
    static_assert(!LS || P == 32, "sector input feeds 32-byte pieces");
    constexpr int SK = LS ? 64 : P;                                  // alignment of the stream coordinates (what the loads are aligned to)
    constexpr int NW = 4 + P / 4, NL = LS ? 4 : P / 16;              // window dwords, loads per request
    constexpr int HPR = FU / 16, RECS_PER_STORE = 64 / HPR;          // helper lanes per flush record, records per store instruction
    constexpr int kFlushRecs = (LZ4HIP_DEC4_FLUSH_RECS) ? (LZ4HIP_DEC4_FLUSH_RECS) : FS * RECS_PER_STORE;
    constexpr bool kLineNoWrap = R % 64 == 0;                        // a 64-byte line of the ring (16 rows from a multiple of 16) never wraps inside
    constexpr int kNearMax = R - 20;                                 // an append writes whole dwords, up to 19 bytes past its last byte
    // vector-memory instructions per iteration: FS flush stores (if it flushes), far fetch, NL input loads (if it requests input)
    // Flush record i of this wavefront.  The 512 bytes of records per wavefront lie behind the ring as TWO halves, one in each of the first two
    // "rows" past the ring's end, at this wavefront's lanes' columns: an append's rows that run past the last ring row land in the storing
    // wavefront's OWN records (rewritten before they are next read) or outside the allocation, never in a neighbour's.  (WAVES == 1: lds + ring + 16 i.)
    unsigned char* const frec_base = lds + kRingBytes + (uint32_t)wave * 256u;
    auto frec = [&](int i) -> Aligned16* {
        if (kFlushRecs <= 16) return (Aligned16*)(frec_base + (uint32_t)i * 16u);                    // (one half suffices: the default configuration)
        return (Aligned16*)(frec_base + (uint32_t)(i >> 4) * kRow + (uint32_t)(i & 15) * 16u);
    };
    const uint32_t lane4 = (uint32_t)(wave * 64 + lane) << 2;

    // ring-relative byte address (row * 256 | lane * 4) plus k rows, wrapped
    auto ring_add = [](uint32_t a, uint32_t rows256) -> uint32_t {
        const uint32_t t = a + rows256;
        if (RPOW2) return t & (kRingBytes - 1u);
        const uint32_t u = t - kRingBytes;                           // (underflows unless t ran past the last row)
        return u < t ? u : t;
    };
#define L4_RING(a) (*(uint32_t*)(lds + (a)))
#define L4_PHASE_SEL(p_) wv::alignbyte(0x07060504u, 0x03020100u, (uint32_t)(p_) & 3u)

    // ---- per-lane state ----
    int skew = 0, in_total = 0;
    uint64_t src_al = 0;
    uint32_t out_limit = 0;      // lz4.c:893 / :1024: a match may not end past oend - LASTLITERALS
    int ip = 0;                  // input cursor (block coordinates): next token / next streamed literal / next header
    // input window: W holds the bytes [wb, wb + 16 + P) of the aligned stream (wb = -16 mod P), L the piece that follows
    uint32_t W[NW];
    wv::u32x4 L[NL];
    int wb = -16;
    int lvalid = 0;              // L holds the piece at wb + 16 + P
    int pend_a = 0, pend_b = 0;  // the loads of that piece are in flight (by the parity of the iteration that issued them)
    int op = 0, fl = 0;          // bytes produced / bytes stored to dst (multiple of FU until the end of the block)
    uint32_t oa = lane4;         // ring address of the dword that contains op
    int kind = kK4None, rem = 0, off = 8;
    int gready = 0;              // kK4Far: the 16 bytes fetched in the previous iteration are this lane's next chunk
    // parsed-ahead sequence
    int pv = 0, p_ll = 0, p_st = 0, p_ml = 0, p_off = 0, p_flags = 0, p_res = 0;
    uint32_t p_l0 = 0, p_l1 = 0, p_l2 = 0;
    int hdr = 0;                 // the cursor is at a sequence's offset field (its literals were streamed)
    uint32_t token = 0;
    int final_seen = 0, final_run = 0, result = 0, done = 0;
    int exhausted = 0;           // (persistent kernels) this lane has no further block
    wv::u32x4 fa = { 0, 0, 0, 0 }, fb = { 0, 0, 0, 0 };
#if defined(LZ4HIP_DEC4_BALLAST_VALU) || defined(LZ4HIP_DEC4_BALLAST_LDS)
    uint32_t ballast = 0;
#endif
#pragma unroll
    for (int j = 0; j < NW; j++) W[j] = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) L[j] = wv::u32x4{ 0, 0, 0, 0 };
    // (Re)start this lane on the block src / iend / dst / oend: every piece of per-block state.  The window starts EMPTY, one piece before the
    // stream (wb = -16 - P): the first trips of the loop request piece 0 into L, take it into W and request piece 1 like every later
    // piece -- no loads here, so a lane of a persistent kernel can restart while its wavefront's accesses are in flight (a piece of the
    // finished block that lands in L afterwards is overwritten by the new block's piece 0, which was issued later: loads return in order).
    auto start_block = [&](bool act) {
        skew = (int)((uint64_t)src & (uint64_t)(SK - 1));
        src_al = (uint64_t)src - (uint64_t)skew;
        in_total = iend > 0 ? (int)(((int64_t)skew + iend + P - 1) & ~(int64_t)(P - 1)) : 0;
        out_limit = oend > kLastLiterals ? (uint32_t)(oend - kLastLiterals) : 0u;
        ip = 0; wb = -16 - P; lvalid = 0; pend_a = 0; pend_b = 0; op = 0; fl = 0; oa = lane4;
        kind = kK4None; rem = 0; off = 8; gready = 0;
        pv = 0; p_ll = 0; p_st = 0; p_ml = 0; p_off = 0; p_flags = 0; p_res = 0; hdr = 0; token = 0;
        final_seen = 0; final_run = 0; result = 0; done = 0;
        if (!act || (!KNOWN && iend == 0)) { done = 1; final_seen = 1; }   // lz4.c:946 returns -(0)
    };
    start_block(active);

    // Append the low n_ bytes of the data dwords at op: rotated to the byte phase of op (one v_perm_b32 per dword), the
    // first dword merged into the ring under a byte mask, the others stored whole.
#define L4_STORE2(OFF_, v_) do { const uint32_t w_ = (v_); wv::lds_store_drop<OFF_>(lds, kLdsBytes, oa, w_); wv::lds_store_drop<OFF_>(lds, kLdsBytes, ob_, w_); } while (0)
#define L4_APPEND(d0_, d1_, d2_, d3_, n_, FOUR_)                                                         \
    do {                                                                                                \
        const uint32_t sb_ = (uint32_t)op & 3u;                                                         \
        const uint32_t s_ = wv::alignbyte(0x08070605u, 0x04030201u, sb_ ^ 3u);                          \
        wv::lds_mskor(&L4_RING(oa), 0xFFFFFFFFu << (8u * sb_), wv::perm(d0_, 0u, s_));                  \
        if (kDual) {                                                                                    \
            /* rows oa + 1 .. oa + 4, not wrapped: the rows past the end of the ring land in the flush records (rewritten before \
               they are next read) or outside the allocation (dropped), and the same rows seen from one ring size below land \
               where they belong or below address 0 (dropped) */                                        \
            const uint32_t ob_ = oa - kRingBytes;                                                       \
            const uint32_t w1_ = wv::perm(d1_, d0_, s_), w2_ = wv::perm(d2_, d1_, s_);                  \
            if (kDual == 2) { /* two rows per LDS instruction */                                        \
                wv::lds_store2_rows_drop<1 * WAVES, 2 * WAVES>(lds, kLdsBytes, oa, w1_, w2_);           \
                wv::lds_store2_rows_drop<1 * WAVES, 2 * WAVES>(lds, kLdsBytes, ob_, w1_, w2_);          \
                if (FOUR_) {                                                                            \
                    const uint32_t w3_ = wv::perm(d3_, d2_, s_), w4_ = wv::perm(0u, d3_, s_);           \
                    wv::lds_store2_rows_drop<3 * WAVES, 4 * WAVES>(lds, kLdsBytes, oa, w3_, w4_);       \
                    wv::lds_store2_rows_drop<3 * WAVES, 4 * WAVES>(lds, kLdsBytes, ob_, w3_, w4_);      \
                } else {                                                                                \
                    L4_STORE2(3 * kRow, wv::perm(0u, d2_, s_));                                         \
                }                                                                                       \
            } else {                                                                                    \
                L4_STORE2(kRow, w1_);                                                                   \
                L4_STORE2(2 * kRow, w2_);                                                               \
                if (FOUR_) {                                                                            \
                    L4_STORE2(3 * kRow, wv::perm(d3_, d2_, s_));                                        \
                    L4_STORE2(4 * kRow, wv::perm(0u, d3_, s_));                                         \
                } else {                                                                                \
                    L4_STORE2(3 * kRow, wv::perm(0u, d2_, s_));                                         \
                }                                                                                       \
            }                                                                                           \
        } else {                                                                                        \
            const uint32_t a1_ = ring_add(oa, kRow), a2_ = ring_add(oa, 2 * kRow), a3_ = ring_add(oa, 3 * kRow); \
            L4_RING(a1_) = wv::perm(d1_, d0_, s_);                                                      \
            L4_RING(a2_) = wv::perm(d2_, d1_, s_);                                                      \
            if (FOUR_) {                                                                                \
                L4_RING(a3_) = wv::perm(d3_, d2_, s_);                                                  \
                L4_RING(ring_add(oa, 4 * kRow)) = wv::perm(0u, d3_, s_);                                \
            } else {                                                                                    \
                L4_RING(a3_) = wv::perm(0u, d2_, s_);                                                   \
            }                                                                                           \
        }                                                                                               \
        oa = ring_add(oa, (((sb_ + (uint32_t)(n_)) << 6) & 0x700u) * (uint32_t)WAVES);                  \
        op += (n_);                                                                                     \
    } while (0)

    // One iteration.  ldF: the far-match registers loaded in THIS iteration; usF: those loaded in the previous one (consumed
    // at the bottom of this one).  ld_pend / us_pend: this lane requested its next piece in this / the previous iteration.
    auto iteration = [&](auto flush_tag, wv::u32x4& ldF, wv::u32x4& usF, int& ld_pend, int& us_pend) __attribute__((always_inline)) -> bool {
        constexpr bool FLUSH = decltype(flush_tag)::value;           // (with FE == 2 only every second iteration flushes)
        constexpr bool INPUT = IE == 1 || !FLUSH;
        constexpr int kVm = (FLUSH ? FS : 0) + 1 + (INPUT ? NL : 0);
        LZ4HIP_ITERATION_HOOK(lane);
        // ================================ TOP ================================
        // ---- (T1) the input window: take the next piece in once the cursor has left the current one; the 16 bytes at the
        //      cursor; the 16 bytes at the source of the current near match ----
        LZ4HIP_SECTION("T1a window refill");
        int d = ip + skew - wb;                                      // byte offset of the cursor in W: 0 .. 16 + P (+ 16 while a piece is awaited)
        {
            const bool no_more = wb + 16 + P >= in_total;            // W holds the last piece of the source
            const bool cross = (d >= P) & ((lvalid != 0) | no_more);
            const wv::mask_t cm = wv::cond(d >= P) & (wv::cond(lvalid != 0) | wv::cond(no_more));   // (lane masks combined by scalar instructions)
#pragma unroll
            for (int j = 0; j < 4; j++) W[j] = wv::sel(cm, W[P / 4 + j], W[j]);
            if constexpr (LS) {
                // the piece at wb + 16 + P is the low or the high half of the sector in L; after the low half L stays valid
                const bool hi_half = ((wb + 16 + P) & 32) != 0;
                const wv::mask_t hm = wv::cond(((wb + 16 + P) & 32) != 0);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    W[4 + 4 * j] = wv::sel(cm, wv::sel(hm, L[2 + j].x, L[j].x), W[4 + 4 * j]); W[5 + 4 * j] = wv::sel(cm, wv::sel(hm, L[2 + j].y, L[j].y), W[5 + 4 * j]);
                    W[6 + 4 * j] = wv::sel(cm, wv::sel(hm, L[2 + j].z, L[j].z), W[6 + 4 * j]); W[7 + 4 * j] = wv::sel(cm, wv::sel(hm, L[2 + j].w, L[j].w), W[7 + 4 * j]);
                }
                lvalid = (cross & hi_half) ? 0 : lvalid;
            } else {
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    W[4 + 4 * j] = wv::sel(cm, L[j].x, W[4 + 4 * j]); W[5 + 4 * j] = wv::sel(cm, L[j].y, W[5 + 4 * j]);
                    W[6 + 4 * j] = wv::sel(cm, L[j].z, W[6 + 4 * j]); W[7 + 4 * j] = wv::sel(cm, L[j].w, W[7 + 4 * j]);
                }
                lvalid = cross ? 0 : lvalid;
            }
            wb += cross ? P : 0;
            d -= cross ? P : 0;
        }
        LZ4HIP_SECTION("T1b cursor view (select tree + rotation)");
        const bool staged16 = d < P;                                 // the 16 bytes at the cursor lie in W (what lies past the source is never used)
        LZ4HIP_STAT(0, true); LZ4HIP_STAT(1, done == 0); LZ4HIP_STAT(2, (done == 0) & !staged16);
        uint32_t x0, x1, x2, x3;
        {
            // W[k .. k + 4], k = d / 4, through a binary tree of selects; then the byte rotation
            const uint32_t k = (uint32_t)d >> 2;
            uint32_t t[NW];
#pragma unroll
            for (int j = 0; j < NW; j++) t[j] = W[j];
            int n = NW;
#pragma unroll
            for (int sh = P / 8; sh >= 1; sh >>= 1) {
                const wv::mask_t m = wv::cond((k & (uint32_t)sh) != 0u);
                n -= sh;
#pragma unroll
                for (int j = 0; j < NW; j++)
                    if (j < n) t[j] = wv::sel(m, t[j + sh], t[j]);
            }
            const uint32_t sx = L4_PHASE_SEL(d);
            x0 = wv::perm(t[1], t[0], sx); x1 = wv::perm(t[2], t[1], sx); x2 = wv::perm(t[3], t[2], sx); x3 = wv::perm(t[4], t[3], sx);
        }
        LZ4HIP_SECTION("T1c near-match source rows");
        uint32_t v0, v1, v2, v3;
        {
            // row of output byte op - off: (op >> 2) - ((off - (op & 3) + 3) >> 2) rows back from oa
            const uint32_t offn = kind == kK4Near ? (uint32_t)off : 4u;        // (any other kind: some valid row)
            const uint32_t back = (((offn + 3u - ((uint32_t)op & 3u)) << 6) & ~0xFFu) * (uint32_t)WAVES;
            uint32_t sa;
            if (RPOW2) sa = (oa - back) & (kRingBytes - 1u);
            else { const uint32_t t = oa - back, u = t + kRingBytes; sa = u < t ? u : t; }
            const uint32_t r0 = L4_RING(sa), r1 = L4_RING(ring_add(sa, kRow)), r2 = L4_RING(ring_add(sa, 2 * kRow)),
                           r3 = L4_RING(ring_add(sa, 3 * kRow)), r4 = L4_RING(ring_add(sa, 4 * kRow));
            const uint32_t sr = L4_PHASE_SEL((uint32_t)op - offn);
            v0 = wv::perm(r1, r0, sr); v1 = wv::perm(r2, r1, sr); v2 = wv::perm(r3, r2, sr); v3 = wv::perm(r4, r3, sr);
        }

        LZ4HIP_SECTION("T2 cooperative flush");
        // ---- (T2) flush finished output, FU bytes at a time, FU / 16 lanes per unit: ALWAYS FS store instructions ----
        if constexpr (FLUSH) {
            const bool need = (done == 0) & (op - fl >= FU);
            const uint64_t needy = wv::ballot(need);
            const int cnt_all = wv::popc64(needy);
            const bool go = cnt_all != 0;                            // wave-uniform
            int cnt = 0;
            bool mine = false;
            if (go) {
                cnt = cnt_all < kFlushRecs ? cnt_all : kFlushRecs;
                const int frank = wv::rank_below(needy);
                mine = need & (frank < kFlushRecs);
                if (mine) {
                    const uint64_t dp = (uint64_t)dst;
                    // ring address of the unit: fl is a multiple of 4, so it starts floor((op - fl) / 4) rows before oa's row
                    *frec(frank) = Aligned16{ { ring_add(oa, kRingBytes - ((((uint32_t)(op - fl)) << 6) & ~0xFFu) * (uint32_t)WAVES), (uint32_t)fl, (uint32_t)dp, (uint32_t)(dp >> 32) } };
                }
                wv::mem_sync();
            }
            const int sub = lane % HPR;
#pragma unroll
            for (int base = 0; base < FS * RECS_PER_STORE; base += RECS_PER_STORE) {
                const int idx = base + lane / HPR;
                const bool act = idx < cnt;
                const wv::mask_t act_m = wv::cond(idx < cnt);
                uint64_t g = 0;
                uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                if (act) {
                    const Aligned16 r = *frec(idx);
                    g = ((uint64_t)r.w[2] | ((uint64_t)r.w[3] << 32)) + (uint64_t)(r.w[1] + 16u * (uint32_t)sub);
                    // the owner's lane bits are in r.w[0]; this helper takes rows 4*sub .. 4*sub+3 of the unit
                    const uint32_t b0 = ring_add(r.w[0], 4u * kRow * (uint32_t)sub);
                    if (kLineNoWrap) { q0 = L4_RING(b0); q1 = L4_RING(b0 + kRow); q2 = L4_RING(b0 + 2 * kRow); q3 = L4_RING(b0 + 3 * kRow); }   // (immediate offsets)
                    else { q0 = L4_RING(b0); q1 = L4_RING(ring_add(b0, kRow)); q2 = L4_RING(ring_add(b0, 2 * kRow)); q3 = L4_RING(ring_add(b0, 3 * kRow)); }
                }
                wv::vm_store16_mask(act_m, g, q0, q1, q2, q3);
            }
            LZ4HIP_STAT(3, need); LZ4HIP_STAT(4, need & !mine);
            if (go) {
                wv::mem_sync();                                      // records and ring rows are free to be overwritten again
                fl += mine ? FU : 0;
            }
        }
        LZ4HIP_SECTION("T3 chunk size");
        // ---- (T3) size of this iteration's chunk of the current copy (appended at the bottom) ----
        const bool room = op - fl <= R - 46;                         // this iteration's appends (<= 16 + 11 bytes + 19 of overshoot) stay clear of unflushed output
        const bool near = kind == kK4Near, lit = kind == kK4Lit;
        const bool can = room & (rem > 0) & !((kind == kK4Far) & (gready == 0)) & !(lit & !staged16);
        const int stride = (near & (off < 16)) ? off : 16;          // a near match whose source would overlap the chunk copies `off` bytes and doubles off
        int n = can ? (rem < stride ? rem : stride) : 0;
        n = ((kind == kK4Zero) & (n > 8)) ? 8 : n;
        LZ4HIP_STAT(5, (done == 0) & (rem > 0)); LZ4HIP_STAT(6, (done == 0) & (rem > 0) & (n == 0)); LZ4HIP_STAT(7, (done == 0) & (rem > 0) & !room);
        LZ4HIP_STAT(8, (done == 0) & (rem > 0) & (kind == kK4Far) & (gready == 0)); LZ4HIP_STAT(9, (done == 0) & (rem > n));
        const int rem_after = rem - n;
        const int op_end = op + rem;                                 // where the current copy ends = where the parsed-ahead sequence's literals go

        LZ4HIP_SECTION("T4 parse ahead");
        // ---- (T4) parse ahead: the next sequence's header (needs only the cursor) ----
        const bool may_parse = (pv == 0) & (final_seen == 0) & !(lit & (rem > 0)) & staged16;
        LZ4HIP_STAT(10, may_parse); LZ4HIP_STAT(11, (done == 0) & (pv != 0));
        if (may_parse) {
            const uint32_t tok = hdr ? token : (x0 & 255u);
            const uint32_t t4 = tok >> 4, b1 = (x0 >> 8) & 255u, mlc = tok & 15u;
            const bool e1 = (hdr == 0) & (t4 == 15u), e2 = mlc == 15u;
            const int ll = hdr ? 0 : (int)t4 + (e1 ? (int)b1 : 0);
            const bool in_win = (hdr != 0) | (t4 <= 11u);             // literals (<= 11, no length byte), offset and first match-length byte are within 16 bytes
            const int o = hdr ? 0 : 1 + (int)t4;                      // position of the offset field when in_win (0 .. 12)
            // offset + first match-length byte: bytes o .. o + 2 of the 16-byte view
            uint32_t ot;
            {
                const uint32_t q = (uint32_t)o >> 2;
                const wv::mask_t m1 = wv::cond((q & 1u) != 0u), m2 = wv::cond((q & 2u) != 0u);
                const uint32_t lo = wv::sel(m2, wv::sel(m1, x3, x2), wv::sel(m1, x1, x0));
                const uint32_t hi = wv::sel(m2, x3, wv::sel(m1, x2, x1));     // (q == 3: o == 12, byte phase 0, hi is not used)
                ot = wv::alignbyte(hi, lo, (uint32_t)o & 3u);
            }
            const int vo = (int)(ot & 0xFFFFu);
            const uint32_t extb = (ot >> 16) & 255u;
            const int ml = (int)mlc + kMinMatch + (e2 ? (int)extb : 0);
            const int lit_end = op_end + ll;
            const int p_after = ip + o + 2;                          // after the offset field
            // what the 16-byte view cannot decide goes to the byte-wise parser: the end of the source, length bytes of 255,
            // the final literal run (lz4.c:851 / :965), every error (lz4.c:863,893 / :980,1024)
            bool trap = (ip + 16 > iend) | (e1 & (b1 == 255u)) | (ip + 1 + (e1 ? 1 : 0) + ll > iend);
            trap |= in_win & ((e2 & (extb == 255u)) | (vo > lit_end) | ((uint32_t)lit_end + (uint32_t)ml > out_limit));   // (lit_end < 2^31 + 270, ml <= 274: no wrap)
            if (KNOWN) trap |= (hdr == 0) & (lit_end > oend - 8);
            else       trap |= ((hdr == 0) & ((lit_end > oend - kMfLimit) | (ip + 1 + (e1 ? 1 : 0) + ll > iend - 8))) | (in_win & e2 & !(p_after < iend - (kLastLiterals + 1)));
            p_l0 = wv::alignbyte(x1, x0, 1); p_l1 = wv::alignbyte(x2, x1, 1); p_l2 = wv::alignbyte(x3, x2, 1);
            token = tok;
            p_ll = in_win ? ll : 0;
            p_st = in_win ? 0 : ll;
            p_ml = in_win ? ml : 0;
            p_off = vo;
            p_flags = in_win ? 0 : (int)kF4Header;
            const int ip_fast = in_win ? p_after + (e2 ? 1 : 0) : ip + 1 + (e1 ? 1 : 0);
            if (trap) {
                LZ4HIP_SECTION("T4x byte-wise parser (trap)");
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
                        p_flags = kF4Final;
                        p_res = KNOWN ? pos + l : le;
                        final_seen = 1;
                    } else {
                        if (KNOWN && err == 0 && pos + l > iend) err = -pos;     // never read literals past the source
                        p_flags = kF4Header;
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
                if (err != 0) { p_flags = kF4Err; p_res = err; }
                hdr = (p_flags & kF4Header) ? 1 : 0;
            } else {
                ip = ip_fast;
                hdr = in_win ? 0 : 1;
            }
            LZ4HIP_SECTION("T4 parse ahead");
            pv = 1;
        }

        LZ4HIP_SECTION("T5 far fetch");
        // ---- (T5) far fetch for the chunk appended at the bottom of the NEXT iteration: ALWAYS one load instruction ----
        // continuation of the current far match, or the first 16 bytes of the parsed-ahead match if the current copy ends
        // in this iteration (the source of the chunk appended at output position p is p - off; a lane that could not append
        // what it holds simply fetches the same bytes again)
        {
            // (the lane masks of the simple comparisons, combined by scalar instructions)
            const wv::mask_t m_rem0 = wv::cond(rem_after == 0);
            const wv::mask_t f_cont = wv::cond(kind == kK4Far) & ~m_rem0;
            const wv::mask_t f_first = m_rem0 & wv::cond(op - fl <= R - 46) & wv::cond(pv != 0) & wv::cond(p_ml != 0) & wv::cond(p_flags == 0) & wv::cond(p_off > kNearMax);
            const int f_pos = (int)wv::sel(f_cont, (uint32_t)(op + n - off), (uint32_t)(op_end + p_ll - p_off));
            const wv::mask_t f_want = f_cont | f_first;
            const wv::mask_t f_do = f_want & wv::cond(f_pos + 16 <= fl);
            wv::vm_load16_mask<POL & 3>(f_do, (uint64_t)dst + (uint64_t)(uint32_t)f_pos, ldF);
            gready = (int)wv::sel(f_do, 1u, 0u);                     // (only read while kind == kK4Far)
            LZ4HIP_STAT(15, gready != 0); LZ4HIP_STAT(16, wv::sel(f_want & ~f_do, 1u, 0u) != 0u);
        }

        LZ4HIP_SECTION("T6 input request");
        // ---- (T6) the next piece of input, once L is free: ALWAYS NL load instructions, each lane for itself ----
        if constexpr (INPUT) {
            const int lpos = wb + 16 + P;                            // aligned stream position of the piece L is for (sector input: a multiple of 64 whenever L is free)
            const wv::mask_t req = wv::cond((done | lvalid | us_pend) == 0) & wv::cond(lpos < in_total);
            const uint64_t g = src_al + (uint64_t)(uint32_t)lpos;
            wv::vm_load16_mask_off<(POL >> 2) & 3, 0>(req, g, L[0]);
            wv::vm_load16_mask_off<(POL >> 2) & 3, 16>(req, g, L[1]);
            if constexpr (NL == 4) {
                wv::vm_load16_mask_off<(POL >> 2) & 3, 32>(req, g, L[2]);
                wv::vm_load16_mask_off<(POL >> 2) & 3, 48>(req, g, L[3]);
            }
            ld_pend = (int)wv::sel(req, 1u, 0u);
        } else {
            ld_pend = 0;
        }

        // ================================ BOTTOM ================================
        LZ4HIP_SECTION("B1 wait");
        // ---- (B1) the loads of the PREVIOUS iteration have landed (this iteration's kVm accesses stay in flight) ----
        wv::vm_wait_list<kVm>(usF, L);
        lvalid = us_pend ? 1 : lvalid;
        us_pend = 0;

#if defined(LZ4HIP_DEC4_BALLAST_VALU) || defined(LZ4HIP_DEC4_BALLAST_LDS)
        // (diagnostic builds only, tools/r04/call16.sh: what does the iteration rate do with N more vector-ALU / LDS instructions?)
#ifdef LZ4HIP_DEC4_BALLAST_VALU
#pragma unroll
        for (int bi = 0; bi < LZ4HIP_DEC4_BALLAST_VALU; bi++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ballast) : "v"(lane4));
#endif
#ifdef LZ4HIP_DEC4_BALLAST_LDS
#pragma unroll
        for (int bi = 0; bi < LZ4HIP_DEC4_BALLAST_LDS; bi++) wv::lds_store_drop<0>(lds, kLdsBytes, 0x80000000u + lane4, ballast);
#endif
#endif
        LZ4HIP_SECTION("B3 chunk append");
        // ---- (B3) append the chunk ----
        {
            const bool far_src = kind == kK4Far;
            v0 = lit ? x0 : (far_src ? usF.x : v0);
            v1 = lit ? x1 : (far_src ? usF.y : v1);
            v2 = lit ? x2 : (far_src ? usF.z : v2);
            v3 = lit ? x3 : (far_src ? usF.w : v3);
            if ((kind == kK4Zero) & (n > 0)) {                       // offset 0 (corrupt streams only): keep what dst holds, 8 bytes at a time
                uint64_t acc = 0;
                for (int b = 0; b < n; b++) acc |= (uint64_t)dst[op + b] << (8 * b);
                v0 = (uint32_t)acc; v1 = (uint32_t)(acc >> 32);
            }
            ip += lit ? n : 0;
            L4_APPEND(v0, v1, v2, v3, n, true);
            rem = rem_after;
            off = (near & (off < 16) & (n > 0)) ? off * 2 : off;
            kind = rem == 0 ? (int)kK4None : kind;
        }

        LZ4HIP_SECTION("B4 promote + literal append");
        // ---- (B4) promote the parsed-ahead sequence: its inline literals, then its copy becomes the current one ----
        {
            const bool promote = (rem == 0) & (pv != 0) & room;
            LZ4HIP_STAT(12, promote); LZ4HIP_STAT(13, (done == 0) & (rem == 0) & (pv != 0) & !room); LZ4HIP_STAT(14, (done == 0) & (rem == 0) & (pv == 0));
            const bool perr = promote & ((p_flags & kF4Err) != 0);
            const bool pgo = promote & !perr;
            L4_APPEND(p_l0, p_l1, p_l2, 0u, pgo ? p_ll : 0, false);
            if (perr) {                                              // corrupt stream: this lane is finished, nothing more is stored
                done = 1; final_seen = 1; final_run = 0; result = p_res;
            }
            if (pgo) {
                const bool streamed = p_st > 0;
                rem = streamed ? p_st : p_ml;
                off = streamed ? off : p_off;
                kind = streamed ? (int)kK4Lit : (p_ml == 0 ? (int)kK4None : (p_off == 0 ? (int)kK4Zero : (p_off <= kNearMax ? (int)kK4Near : (int)kK4Far)));
                final_run = (p_flags & kF4Final) ? 1 : final_run;
                result = (p_flags & kF4Final) ? p_res : result;
            }
            pv = promote ? 0 : pv;
        }

        LZ4HIP_SECTION("B5 end of block + loop");
        // ---- (B5) end of block: write out the last bytes exactly ----
        if (final_run && rem == 0 && !pv && !done) {
            uint32_t qa = ring_add(oa, kRingBytes - ((((uint32_t)(op - fl)) << 6) & ~0xFFu) * (uint32_t)WAVES);
            while (op - fl >= 4) { const uint32_t qd = L4_RING(qa); __builtin_memcpy(dst + fl, &qd, 4); fl += 4; qa = ring_add(qa, kRow); }
            if (fl < op) {
                const uint32_t qd = L4_RING(qa);
                for (int b = 0; fl + b < op; b++) dst[fl + b] = (uint8_t)(qd >> (8 * b));
            }
            done = 1;
        }
        return !wv::any(done == 0);                                  // every lane of the wavefront is finished
    };

    for (;;) {
        if (!std::remove_reference<NEXT>::type::kPersistent) {
            if (iteration(std::true_type{}, fa, fb, pend_a, pend_b)) break;
            if (iteration(std::integral_constant<bool, FE == 1>{}, fb, fa, pend_b, pend_a)) break;
        } else {
            (void)iteration(std::true_type{}, fa, fb, pend_a, pend_b);
            (void)iteration(std::integral_constant<bool, FE == 1>{}, fb, fa, pend_b, pend_a);
            // lanes that have finished their block take the next one (a rare, wave-uniform branch: once per block and lane)
            const bool fin = (done != 0) & (exhausted == 0);
            if (wv::any(fin)) {
                if (fin) {
                    if (next(result, src, iend, dst, oend)) start_block(true);
                    else exhausted = 1;
                }
                if (!wv::any(exhausted == 0)) break;
            }
        }
    }
    return result;
#undef L4_RING
#undef L4_PHASE_SEL
#undef L4_APPEND
#undef L4_STORE2
}

// How many blocks of the batch the filter selects (one launch in front of a large partitioned batch): the two lane kernels below
// look at this number and only ONE of them runs -- one block per lane when (nearly) every block is selected, the persistent grid
// when many are not (its lanes skip those instead of idling through a whole wavefront's lifetime).
__global__ void __launch_bounds__(256) count_selected_kernel(Batch b, int filter, unsigned* count)
{
    unsigned mine = 0;
    for (int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x; blk < b.n_blocks; blk += (int64_t)gridDim.x * 256)
        mine += block_selected(filter, batch_src_len(b, blk), batch_dst_cap(b, blk)) ? 1u : 0u;
    const unsigned total = wv::scan_add(mine);                       // (inclusive prefix sum: the last lane holds the wavefront's sum)
    if ((threadIdx.x & 63u) == 63u && total) atomicAdd(count, total);
}
// gate_mode 0: run; 1: run only if *gate >= threshold; 2: run only if *gate < threshold
LZ4HIP_DEVICE bool lane4_gate_open(const unsigned* gate, int gate_mode, unsigned threshold)
{
    if (gate_mode == 0) return true;
    const unsigned c = wv::uniform(*gate);
    return gate_mode == 1 ? c >= threshold : c < threshold;
}

// Load-time check of the hardware rule the dual ring stores rest on (L4_APPEND with kDual != 0): a DS store whose address lies outside
// the workgroup's LDS allocation -- past its end, or "below zero" after a 32-bit wrap -- is DROPPED: no fault, and no byte of this or any
// co-resident workgroup's allocation changes.  Every workgroup (one wavefront, the decoder's 12 800 bytes, twelve resident per CU) fills its
// allocation with a pattern of its own, issues exactly the decoder's kinds of out-of-range stores (ds_write_b32 with an offset,
// ds_write2st64_b32 with one or both rows outside, from `row` and from `row - ring size`), lingers so that its neighbours overlap with it,
// and counts the pattern words that changed.  lz4hip_api.hip runs it once per device before the first lane-mapped decode; anything but
// zero selects the wrapped-row instantiation (POL bit 5) for that device.  (tools/lds_out_of_range.hip is the stand-alone, longer form.)
// NOTE: `lds` must be the ONLY __shared__ object of these kernels -- the dual stores assume that the ring starts at LDS address 0.
__global__ void __launch_bounds__(64) lds_drop_probe_kernel(unsigned* errors, int rounds)
{
    constexpr uint32_t kBytes = lane4_lds_bytes(192), kRing = 64u * 192u, kWords = kBytes / 4u;
    LZ4HIP_STATIC_LDS(lds, kBytes);
    uint32_t* const s = (uint32_t*)lds;
    const uint32_t lane = threadIdx.x, tag = 0x9E3779B9u * (blockIdx.x + 1u);
    for (uint32_t i = lane; i < kWords; i += 64u) s[i] = tag ^ i;
    wv::block_sync();
    for (int r = 0; r < rounds; r++) {
        const uint32_t junk = 0xDEAD0000u | (uint32_t)r;
        const uint32_t last = kRing - 256u + lane * 4u;              // the ring's last row: rows + 1 .. + 4 of an append from here
        const uint32_t own = tag ^ (last >> 2);
        // (a) from `row`: the rows past the ring land in the flush records (in range, rewritten below) or past the allocation (dropped)
        wv::lds_store2_rows_drop<0, 3>(lds, kBytes, last, own, junk);      // row 0 in range, row + 3 = 256 bytes past the END of the allocation
        wv::lds_store2_rows_drop<3, 4>(lds, kBytes, last, junk, junk);     // both past the end
        wv::lds_store_drop<1024>(lds, kBytes, last, junk);
        wv::lds_store_drop<768>(lds, kBytes, kBytes - 256u + lane * 4u, junk);
        // (b) from `row - ring size` for rows that did not wrap: "negative" addresses
        const uint32_t neg = lane * 4u - kRing;
        wv::lds_store2_rows_drop<1, 2>(lds, kBytes, neg, junk, junk);
        wv::lds_store2_rows_drop<3, 4>(lds, kBytes, neg + 2048u, junk, junk);
        wv::lds_store_drop<768>(lds, kBytes, neg, junk);
        wv::lds_store2_rows_drop<0, 48>(lds, kBytes, neg, junk, tag ^ lane);   // row 0 below zero (dropped), row 48 = this lane's dword of row 0 (in range)
        // the flush records behind the ring take (a)'s in-range overshoot in the decoder; here they keep their pattern
        wv::mem_sync();
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_sleep(8);
#endif
    }
    wv::block_sync();
    unsigned bad = 0;
    for (uint32_t i = lane; i < kWords; i += 64u) bad += s[i] != (tag ^ i) ? 1u : 0u;
    if (bad) atomicAdd(errors, bad);
}

// One wavefront per workgroup; lane i of workgroup g decodes block g*64 + i.
template <bool KNOWN, int R, int P, int FU, int FS, int FE = 1, int IE = 1, int POL = 0>
__global__ void __launch_bounds__(64) decode_lane4_kernel(Batch b, int filter, const unsigned* gate = nullptr, int gate_mode = 0, unsigned threshold = 0)
{
    LZ4HIP_STATIC_LDS(lds, lane4_lds_bytes(R));
    if (!lane4_gate_open(gate, gate_mode, threshold)) return;
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
    const int r = lane4_decode_block<KNOWN, R, P, FU, FS, FE, IE, POL>(lds, lane, active, src, src_len, dst, out_size);
    if (active) b.result[blk] = r;
}

// FOUR wavefronts per workgroup: for batches that cannot give every SIMD a wavefront of its own anyway.  A wavefront alone on its SIMD needs
// 8.4 ms for its 64 D2 blocks, two on one SIMD 10.5 each -- and where the hardware puts 1 024 single-wavefront workgroups depends on what ran
// before them (a 65 536-block decode: 8.5 ms back to back, 10.5 ms behind the wavefront-mapped launch of the same call:
// profiles/r06/decoder_mid_batches_placement.txt).  The four wavefronts of ONE workgroup go to the four SIMDs of one CU, and as many workgroups
// as there are CUs spread one per CU.  The rings of the four wavefronts are interleaved row by row across all 256 lanes (WAVES = 4 of
// lane4_decode_block), so the ring still starts at LDS address 0 and the dual ring stores keep working.
template <bool KNOWN, int R, int P, int FU, int FS, int FE = 1, int IE = 1, int POL = 0>
__global__ void __launch_bounds__(256) decode_lane4_wg4_kernel(Batch b, int filter)
{
    LZ4HIP_STATIC_LDS(lds_all, 4u * lane4_lds_bytes(R));
    const int lane = (int)(threadIdx.x & 63u);
    const int w = wv::wave_in_block();
    const int64_t blk = ((int64_t)blockIdx.x * 4 + w) * 64 + lane;
    bool active = blk < b.n_blocks;
    int src_len = 0, out_size = 0;
    if (active) {
        src_len = batch_src_len(b, blk); out_size = batch_dst_cap(b, blk);
        active = block_selected(filter, src_len, out_size);
    }
    if (!wv::any(active)) return;
    const uint8_t* src = active ? batch_src(b, blk) : nullptr;
    uint8_t* dst = active ? batch_dst(b, blk) : nullptr;
    const int r = lane4_decode_block<KNOWN, R, P, FU, FS, FE, IE, POL, NoNext, 4>(lds_all, lane, active, src, src_len, dst, out_size, NoNext(), w);
    if (active) b.result[blk] = r;
}

// (what a lane of the persistent kernel does between two blocks)
struct PullNext {
    static constexpr bool kPersistent = true;
    const Batch* b; int filter; unsigned long long* counter; int64_t cur;
    LZ4HIP_DEVICE bool operator()(int r, const uint8_t*& src, int& iend, uint8_t*& dst, int& oend)
    {
        if (cur >= 0) b->result[cur] = r;
        for (;;) {
            const int64_t blk = (int64_t)atomicAdd(counter, 1ull);
            if (blk >= b->n_blocks) { cur = -1; return false; }
            const int sl = batch_src_len(*b, blk), oc = batch_dst_cap(*b, blk);
            if (!block_selected(filter, sl, oc)) continue;
            cur = blk; src = batch_src(*b, blk); iend = sl; dst = batch_dst(*b, blk); oend = oc;
            return true;
        }
    }
};

// Persistent variant: a grid of resident wavefronts whose lanes pull block numbers from `counter` (zeroed by the host) until the
// batch is exhausted; blocks the filter does not select are skipped by the lane that drew them.
template <bool KNOWN, int R, int P, int FU, int FS, int FE = 1, int IE = 1, int POL = 0>
__global__ void __launch_bounds__(64) decode_lane4_persistent_kernel(Batch b, int filter, unsigned long long* counter, const unsigned* gate = nullptr,
                                                                     int gate_mode = 0, unsigned threshold = 0)
{
    LZ4HIP_STATIC_LDS(lds, lane4_lds_bytes(R));
    if (!lane4_gate_open(gate, gate_mode, threshold)) return;
    const int lane = (int)threadIdx.x;
    PullNext next{ &b, filter, counter, -1 };
    const uint8_t* src = nullptr; uint8_t* dst = nullptr; int src_len = 0, out_size = 0;
    const bool active = next(0, src, src_len, dst, out_size);
    if (!wv::any(active)) return;
    (void)lane4_decode_block<KNOWN, R, P, FU, FS, FE, IE, POL, PullNext&>(lds, lane, active, src, src_len, dst, out_size, next);
}

}  // namespace lz4hip
