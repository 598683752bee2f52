// lz4hip_encode_sm.hpp -- batched LZ4 fast block encoder (64k variant) for gfx950, one LANE per
// block, written as a CONVERGENT state machine; bit-exact to LZ4_compress64kCtx
// (original/lz4.c:573-771 == LZ4_compress64kCtx_64, src/LZ4pn/LZ4Codec.Unsafe64.Dirty.cs:303-528).
//
// Why: in the straightforward lane-per-block encoder (lz4hip_encode_lane.hpp) every lane runs the
// reference's nested loops, and a SIMT loop runs until its LAST lane leaves it: the search loop
// averages 1.9 probes per sequence on fuzzer-style data but the slowest of 64 lanes needs ~10, each
// probe being two dependent memory round trips.  Measured: 39 GB/s, unchanged when a third of its
// memory requests were removed -- it is bound by that max-over-lanes serialisation, not by requests.
// Here every lane carries an explicit state and every loop iteration advances every lane by ONE
// memory round trip (consume what was requested in the previous iteration, decide, request the
// next): lanes never wait for each other.
//
//   PROBE_T  table value arrived  -> overwrite bucket, request the candidate's bytes
//   PROBE_R  candidate arrived    -> mismatch: next probe (skip schedule) | match: catch-up, emit
//                                    literal part, request the bytes after the match start
//   COUNT    16 more bytes arrived-> still equal: request the next 16 | done: token, offset, length
//                                    bytes; request the 8 bytes at ip-2
//   POST_W   those arrived        -> re-seed table at ip-2, request the bucket of ip
//   POST_T / POST_R               -> the reference's "test next position" (zero-literal sequence or
//                                    back to the search)
//   LIT / TAIL                    -> literal runs too long to ride in the packed header store
//
// Hash table: u16 positions as in the reference, but tagged with a per-lane block epoch in the upper
// half of a 32-bit entry so that a stale entry of the previous block reads as "empty" (== position 0,
// exactly what the reference's freshly zeroed table yields) and no per-block clearing is needed.
// The slab (32 KiB per resident lane) is zeroed once per launch.  Blocks of 65547 bytes and more
// (generic variant, u32 table + distance check) are left to lz4hip_encode.hpp.
#pragma once
#include "lz4hip_common.hpp"
#include "lz4hip_decode_lane.hpp"   // load_u64 / store_u64
#include "lz4hip_encode_lane.hpp"   // lane_copy, lane_put_length, lane_count_equal

namespace lz4hip {

constexpr int kSmEncodeWavesPerCu = 16;
constexpr int kSmTableBytes = 8192 * 4;     // 8192 x (epoch:16 | position:16)

enum SmState { kSmNextBlock = 0, kSmProbeT, kSmProbeR, kSmCount, kSmPostW, kSmPostT, kSmPostR, kSmLit, kSmTail, kSmCatch };

LZ4HIP_DEVICE uint32_t sm_hash(uint32_t word) { return (word * kGolden) >> 19; }            // lz4.c:569-570

// Persistent grid: every lane pulls block indices from `counter`; blocks >= LZ4_64KLIMIT are skipped
// (result left untouched for the wavefront kernel).  `tables`: kSmTableBytes per lane, zeroed per launch.
__global__ void __launch_bounds__(64) encode_fast_sm_kernel(Batch b, unsigned long long* counter, uint8_t* tables)
{
    uint32_t* table = (uint32_t*)(tables + ((size_t)blockIdx.x * 64 + threadIdx.x) * kSmTableBytes);
    uint32_t epoch = 0;

    // ---- per-block state ----
    const uint8_t* in = nullptr; uint8_t* out = nullptr;
    int64_t blk = -1;
    int n = 0, cap = 0, mflimit = 0, matchlimit = 0;
    int ip = 0, anchor = 0, op = 0;
    int state = kSmNextBlock;
    int attempts = 0, probe = 0, ref = 0;
    uint32_t cur_word = 0, fwd_word = 0, h = 0;
    uint64_t fw = 0; int fw_pos = 0;                 // 8-byte forward window over the input
    int token_at = 0, ll = 0, packed = 0, lit_from = 0, lit_k = 0;
    uint32_t token = 0;
    int count_base = 0;                              // COUNT: bytes already known equal
    // ---- responses (requested in the previous iteration) ----
    uint32_t r_tv = 0;                               // table entry
    uint64_t r_ref8 = 0, r_ip8 = 0;                  // 8 bytes at ref-4 / ip-4 (candidate word + catch-up bytes)
    Vec16 r_a = { { 0, 0, 0, 0 } }, r_b = { { 0, 0, 0, 0 } };   // COUNT: 16 bytes at ip side / ref side; LIT: literal chunk in r_a
    uint64_t r_pm = 0;                               // POST_W: 8 bytes at ip-2
    Vec16 r_l = { { 0, 0, 0, 0 } };                  // the literal bytes of a packed sequence
    Vec16 r_c = { { 0, 0, 0, 0 } }, r_d = { { 0, 0, 0, 0 } };   // LIT / TAIL: literal chunk

#define SM_FWD_WORD(pos) ((uint32_t)(fw >> (8 * ((pos) - fw_pos))))
#define SM_FWD_REFILL(pos) do { if ((pos) - fw_pos > 4 || (pos) < fw_pos) { fw_pos = (pos); fw = load_u64(in + fw_pos); } } while (0)
#define SM_FINISH(value) do { b.result[blk] = (value); state = kSmNextBlock; } while (0)
    // next probe of the search loop (lz4.c:642-654): advances ip/probe by the skip schedule and requests the bucket
#define SM_NEXT_PROBE()                                                                              \
    do {                                                                                             \
        cur_word = fwd_word; h = sm_hash(cur_word);                                                  \
        const int step_ = attempts++ >> 6;                                                           \
        ip = probe; probe = ip + step_;                                                              \
        if (probe > mflimit) { state = kSmTail; lit_k = -1; }                                        \
        else { SM_FWD_REFILL(probe); fwd_word = SM_FWD_WORD(probe); r_tv = table[h]; state = kSmProbeT; } \
    } while (0)
    // start counting the match that begins at (ip, ref): offset is emitted later together with the token
#define SM_START_COUNT()                                                                             \
    do {                                                                                             \
        count_base = 0; state = kSmCount;                                                            \
        if (ip + 4 + 16 <= matchlimit) { r_a = load_v16(in + ip + 4); r_b = load_v16(in + ref + 4); } \
    } while (0)

    // Two-phase clock: a response register is only ever READ in one phase and WRITTEN (requested) in the
    // other -- or read before it is re-requested inside one branch -- so no branch of an iteration has to
    // wait for a load issued earlier in the same iteration by another branch (the compiler cannot know that
    // the lanes of two branches are disjoint).  The natural order of the parse alternates anyway:
    //   even: PROBE_T / POST_T (read r_tv, request r_ref8/r_ip8), COUNT (read r_a/r_b/r_l, request r_a/r_b or r_pm)
    //   odd : PROBE_R / POST_R (read r_ref8/r_ip8, request r_tv or r_a/r_b/r_l), POST_W (read r_pm, request r_tv), next block,
    //         byte-wise catch-up;   either: LIT / TAIL (own registers)
    // A lane whose state belongs to the other phase idles for one iteration (rare: long matches, long literals).
    for (unsigned iter = 1;; iter++) {
        const bool even = (iter & 1u) == 0u;
        const bool even_state = state == kSmProbeT || state == kSmPostT || state == kSmCount;
        const bool any_phase = state == kSmLit || state == kSmTail;        // own response registers (r_c / r_d)
        if (!any_phase && even != even_state) continue;
        if (state == kSmNextBlock) {
            blk = (int64_t)atomicAdd(counter, 1ull);
            if (blk >= b.n_blocks) return;
            n = batch_src_len(b, blk); cap = batch_dst_cap(b, blk);
            if (n >= k64kLimit) continue;                             // generic variant: not this kernel's job
            in = batch_src(b, blk); out = batch_dst(b, blk);
            mflimit = n - kMfLimit; matchlimit = n - kLastLiterals;
            ip = 0; anchor = 0; op = 0;
            if (++epoch == 0xFFFFu) {                               // epoch tags exhausted: start over on a clean table
                for (int k = 0; k < kSmTableBytes; k += 16) store_v16((uint8_t*)table + k, Vec16{ { 0, 0, 0, 0 } });
                epoch = 1;
            }
            if (n < kMinLength) { state = kSmTail; lit_k = -1; continue; }   // lz4.c:615
            ip = 1;                                                   // lz4.c:631: position 0 is never probed
            fw_pos = ip; fw = load_u64(in + ip);
            fwd_word = SM_FWD_WORD(ip);
            attempts = 67; probe = ip;
            SM_NEXT_PROBE();
            continue;
        }

        if (state == kSmProbeT || state == kSmPostT) {
            // ---- bucket value arrived: read-then-overwrite (lz4.c:650-652 / :742-749) ----
            ref = (r_tv >> 16) == epoch ? (int)(r_tv & 0xFFFFu) : 0;  // stale or empty bucket == position 0
            table[h] = (epoch << 16) | ((uint32_t)ip & 0xFFFFu);
            // candidate word, plus the 4 bytes before it and before ip for the catch-up
            if (ref >= 4 && ip >= 4) { r_ref8 = load_u64(in + ref - 4); r_ip8 = load_u64(in + ip - 4); }
            else { r_ref8 = (uint64_t)load_u32(in + ref) << 32; r_ip8 = ~r_ref8 & 0xFFFFFFFFull; }   // no catch-up bytes: make them differ
            state = state == kSmProbeT ? kSmProbeR : kSmPostR;
        } else if (state == kSmProbeR || state == kSmPostR) {
            const bool hit = (uint32_t)(r_ref8 >> 32) == cur_word;
            if (!hit) {
                if (state == kSmPostR) {                              // lz4.c:754-755
                    anchor = ip++;
                    SM_FWD_REFILL(ip); fwd_word = SM_FWD_WORD(ip);
                    attempts = 67; probe = ip;
                }
                SM_NEXT_PROBE();
            } else {
                if (state == kSmProbeR) {
                    // ---- catch up (lz4.c:657): trailing equal bytes among the 4 before ip / ref ----
                    const uint32_t x = (uint32_t)r_ref8 ^ (uint32_t)r_ip8;
                    int back = x ? (__builtin_clz(x) >> 3) : 4;
                    const int room = ip - anchor < ref ? ip - anchor : ref;
                    const bool fetched = ref >= 4 && ip >= 4;           // were the 4 bytes before ip / ref requested?
                    back = back < room ? back : room;
                    ip -= back; ref -= back;
                    if (!fetched || (back == 4 && room > 4)) { state = kSmCatch; continue; }   // rare: go on byte-wise
                    goto start_sequence;
                }
                // zero-literal sequence (lz4.c:751): token reserved WITHOUT the literal limit test
                anchor = ip; ll = 0; lit_from = ip;
                token_at = op++; token = 0;
                packed = token_at + 16 <= cap;
                SM_START_COUNT();
            }
        } else if (state == kSmCatch) {
            while (ip > anchor && ref > 0 && in[ip - 1] == in[ref - 1]) { ip--; ref--; }
            goto start_sequence;
        } else if (state == kSmCount) {
            // ---- match length (lz4.c:698-721): 16 bytes per round trip, exact tail ----
            const int a = ip + 4 + count_base;
            int eq;
            bool done = true;
            if (a + 16 <= matchlimit) {
                const uint64_t d0 = (r_a.w[0] ^ r_b.w[0]) | ((uint64_t)(r_a.w[1] ^ r_b.w[1]) << 32);
                const uint64_t d1 = (r_a.w[2] ^ r_b.w[2]) | ((uint64_t)(r_a.w[3] ^ r_b.w[3]) << 32);
                eq = d0 ? (__builtin_ctzll(d0) >> 3) : (d1 ? 8 + (__builtin_ctzll(d1) >> 3) : 16);
                if (eq == 16) {
                    count_base += 16; done = false;
                    if (a + 32 <= matchlimit) { r_a = load_v16(in + a + 16); r_b = load_v16(in + ref + 4 + count_base); }
                }
            } else {
                eq = lane_count_equal(in, a, ref + 4 + count_base, matchlimit);   // last < 16 bytes of the block
            }
            if (done) {
                const int extra = count_base + eq;
                const uint32_t off = (uint32_t)(ip - ref) & 0xFFFFu;
                // ---- offset, token, length bytes (lz4.c:693-733) ----
                if (op + 2 > cap) { SM_FINISH(0); continue; }
                if (!packed) { out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8); }
                op += 2;
                if (op + (extra >> 8) > cap - 6 || (extra >= 15 && op + (extra - 15) / 255 + 1 > cap)) { SM_FINISH(0); continue; }   // lz4.c:728
                token |= extra >= 15 ? 15u : (uint32_t)extra;
                if (packed) {
                    // token, <= 13 literals and the offset leave as ONE 16-byte store (see lz4hip_encode_lane.hpp)
                    uint64_t l0 = 0, l1 = 0;
                    if (ll > 0) { l0 = r_l.w[0] | ((uint64_t)r_l.w[1] << 32); l1 = r_l.w[2] | ((uint64_t)r_l.w[3] << 32); }
                    if (ll < 8) { l0 &= (1ull << (8 * ll)) - 1ull; l1 = 0; }
                    else l1 &= (1ull << (8 * (ll - 8))) - 1ull;
                    uint64_t v0 = (l0 << 8) | token, v1 = (l1 << 8) | (l0 >> 56);
                    const int sh = 8 * (1 + ll);                      // 8 .. 112
                    if (sh < 64) { v0 |= (uint64_t)off << sh; v1 |= sh > 48 ? (uint64_t)off >> (64 - sh) : 0ull; }
                    else v1 |= (uint64_t)off << (sh - 64);
                    const Vec16 o = { { (uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32) } };
                    store_v16(out + token_at, o);
                } else {
                    out[token_at] = (uint8_t)token;
                }
                if (extra >= 15) op += lane_put_length(out + op, extra - 15);
                ip += kMinMatch + extra; anchor = ip;
                if (ip > mflimit) { state = kSmTail; lit_k = -1; continue; }    // lz4.c:736
                r_pm = load_u64(in + ip - 2);                         // words at ip-2, ip and ip+1
                state = kSmPostW;
            }
        } else if (state == kSmPostW) {
            // ---- re-seed at ip-2, then test ip (lz4.c:739-749) ----
            fw = r_pm; fw_pos = ip - 2;
            table[sm_hash(SM_FWD_WORD(ip - 2))] = (epoch << 16) | ((uint32_t)(ip - 2) & 0xFFFFu);
            cur_word = SM_FWD_WORD(ip);
            h = sm_hash(cur_word);
            r_tv = table[h];
            state = kSmPostT;
        } else if (state == kSmLit) {
            // ---- a literal run that does not fit the packed store: 16 bytes per round trip ----
            if (lit_k >= 0) {                                          // chunk requested last time has arrived
                const int m = ll - lit_k < 16 ? ll - lit_k : 16;
                if (m == 16) store_v16(out + op + lit_k, r_c);
                else for (int i = 0; i < m; i++) out[op + lit_k + i] = in[lit_from + lit_k + i];
                lit_k += m;
            } else lit_k = 0;
            if (lit_k < ll) { if (ll - lit_k >= 16) r_c = load_v16(in + lit_from + lit_k); }
            else { op += ll; SM_START_COUNT(); }
        } else {                                                       // kSmTail: last literals (lz4.c:758-767)
            const int run = n - anchor;
            if (lit_k < 0) {
                if (op + run + 1 + (run - 15 + 255) / 255 > cap) { SM_FINISH(0); continue; }    // lz4.c:762
                out[op++] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
                if (run >= 15) op += lane_put_length(out + op, run - 15);
                lit_k = 0;
                if (run >= 16) r_d = load_v16(in + anchor);
            } else {
                const int m = run - lit_k < 16 ? run - lit_k : 16;
                if (m == 16) store_v16(out + op + lit_k, r_d);
                else for (int i = 0; i < m; i++) out[op + lit_k + i] = in[anchor + lit_k + i];
                lit_k += m;
                if (run - lit_k >= 16) r_d = load_v16(in + anchor + lit_k);
            }
            if (lit_k >= run) SM_FINISH(op + run);
        }
        continue;

    start_sequence:
        // ---- literal part of the sequence (lz4.c:660-691); the match at (ip, ref) is counted next ----
        ll = ip - anchor;
        token_at = op++;
        if (op + ll + (ll >> 8) > cap - 8 || (ll >= 15 && op + (ll - 15) / 255 + 1 + ll > cap)) { SM_FINISH(0); continue; }   // lz4.c:663
        token = ll >= 15 ? 0xF0u : (uint32_t)(ll << 4);
        lit_from = anchor;
        packed = ll <= 13 && token_at + 16 <= cap && anchor + 16 <= n;
        if (packed) { if (ll > 0) r_l = load_v16(in + anchor); op += ll; SM_START_COUNT(); }
        else {
            if (ll >= 15) op += lane_put_length(out + op, ll - 15);
            state = kSmLit; lit_k = -1;
        }
    }
#undef SM_FWD_WORD
#undef SM_FWD_REFILL
#undef SM_FINISH
#undef SM_NEXT_PROBE
#undef SM_START_COUNT
}

}  // namespace lz4hip
