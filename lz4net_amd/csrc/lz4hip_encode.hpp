// lz4hip_encode.hpp -- batched LZ4 fast block encoder for gfx950, one wavefront per block,
// bit-exact to the reference.
//
// Replaces LZ4_compress64kCtx (original/lz4.c:573-771 == LZ4_compress64kCtx_64,
// src/LZ4pn/LZ4Codec.Unsafe64.Dirty.cs:303-528) for inputs below LZ4_64KLIMIT and LZ4_compressCtx
// (original/lz4.c:345-562 == LZ4_compressCtx_64, :75-297) above it, behind the dispatcher
// LZ4_compress_limitedOutput (original/lz4.c:774-792 == Encode64, src/LZ4pn/LZ4Codec.Unsafe.cs:275-297).
//
// The greedy parse is a chain of hash-table read-modify-writes whose state depends on every
// position visited, so (to stay bit-exact) its control state is wave-uniform: the 16 KiB hash
// table lives in LDS (8192 x u16 / 4096 x u32, zero-filled per block exactly like the reference's
// per-call table), the probe sequence runs on scalars, and the 64 lanes do the data-parallel parts:
// backward catch-up and forward match counting by ballot + count-trailing-zeros over 4 bytes per
// lane (256 bytes per step), literal copies at 16 B per lane, and length-byte fills.
#pragma once
#include "lz4hip_common.hpp"

// Section timers of the wavefront encoder (tools/enc_wave_sections.hip defines these to s_memtime accumulators; nothing in product builds).
#ifndef LZ4HIP_ENC_T0
#define LZ4HIP_ENC_DECL() ((void)0)
#define LZ4HIP_ENC_T0() ((void)0)
#define LZ4HIP_ENC_T(slot) ((void)0)
#define LZ4HIP_ENC_COUNT(slot, n) ((void)0)
#define LZ4HIP_ENC_FLUSH() ((void)0)
#endif

namespace lz4hip {

template <bool GENERIC> struct FastTable;
template <> struct FastTable<false> {   // 64k variant: U16 HashTable[8192], hash shift 19 (lz4.c:567-570,583)
    typedef uint16_t entry;
    static LZ4HIP_DEVICE uint32_t hash(uint32_t word) { return (word * kGolden) >> 19; }
};
template <> struct FastTable<true> {    // generic variant: U32 HashTable[4096], hash shift 20 (lz4.c:183-185,248)
    typedef uint32_t entry;
    static LZ4HIP_DEVICE uint32_t hash(uint32_t word) { return (word * kGolden) >> 20; }
};

// wave-uniform unaligned dword of the input at wave-uniform position p
LZ4HIP_DEVICE uint32_t input_word(const uint8_t* in, int p) { return wv::uniform(load_u32(in + p)); }

// One lane's share of a 256-byte round of the count below: XOR of its dwords at in[a + off + 4*lane] and in[b + ...],
// with a sentinel at the first byte that must not be counted (1 = "differs at byte 0" where nothing may be counted).
LZ4HIP_DEVICE uint32_t common_length_round(const uint8_t* in, int a, int b, int limit, int off)
{
    const int pa = a + off + wv::lane() * 4;
    int valid = limit - pa;                           // bytes of this lane's dword that may be counted
    valid = valid < 0 ? 0 : (valid > 4 ? 4 : valid);
    uint32_t diff = 1;
    if (valid > 0) {
        diff = load_u32(in + pa) ^ load_u32(in + b + off + wv::lane() * 4);
        if (valid < 4) diff |= 1u << (8 * valid);
    }
    return diff;
}

// Number of equal bytes in[a + i] == in[b + i] for i < limit - a (a > b), counted 4 bytes per lane.
// Equals the reference's 8-byte XOR/ctz loop plus its 4/2/1-byte tails (lz4.c:698-721).
// `first_round`: common_length_round(in, a, b, limit, 0), which the caller may have requested earlier (its loads then
// travel together with those of the catch-up instead of costing a round trip of their own).
LZ4HIP_DEVICE int wave_common_length(const uint8_t* in, int a, int b, int limit, uint32_t first_round)
{
    int total = 0;
    uint32_t diff = first_round;
    for (;;) {
        const uint64_t stop = wv::ballot(diff != 0);
        if (stop) {
            const int first = wv::ctz64(stop);
            const uint32_t d = wv::readlane(diff, first);
            return total + first * 4 + (__builtin_ctz(d) >> 3);
        }
        total += 256;
        diff = common_length_round(in, a, b, limit, total);
    }
}

LZ4HIP_DEVICE int wave_common_length(const uint8_t* in, int a, int b, int limit)
{
    return wave_common_length(in, a, b, limit, common_length_round(in, a, b, limit, 0));
}

// Number of bytes the match can be extended backwards: in[ip-1-i] == in[ref-1-i], i < bound.
LZ4HIP_DEVICE int wave_catch_up(const uint8_t* in, int ip, int ref, int bound)
{
    const int lane = wv::lane();
    int total = 0;
    while (total < bound) {
        const int i = total + lane;
        const bool differs = i >= bound || in[ip - 1 - i] != in[ref - 1 - i];
        const uint64_t stop = wv::ballot(differs);
        if (stop) return total + wv::ctz64(stop);
        total += 64;
    }
    return bound;
}

// length bytes for a literal-run / match length whose nibble saturated: `rest` = length - 15
// -> floor(rest / 255) bytes of 255, then rest % 255.  Returns bytes written.
LZ4HIP_DEVICE int put_length_bytes(uint8_t* out, int rest)
{
    const int n255 = rest / 255;
    wave_fill(out, 255, n255);
    if (wv::lane() == 0) out[n255] = (uint8_t)(rest - n255 * 255);
    return n255 + 1;
}

// Sum of the skip schedule's steps over attempts [0, a): step(t) = t >> 6 (lz4.c:645, skipStrength 6).
LZ4HIP_DEVICE int skip_sum(int a) { const int q = a >> 6, r = a & 63; return 32 * q * (q - 1) + q * r; }

// The match search of one sequence (lz4.c:642-654), 64 probes per step.  The probe positions of a search are a fixed
// arithmetic function of the probe count (attempts start at 67, step = attempts >> 6), so lane i takes the i-th next
// probe: it loads its 4 bytes, hashes them, gathers its table entry and the 4 bytes at that candidate; the first lane
// whose candidate matches (or whose successor would leave the input, lz4.c:648) ends the search.
// What makes the serial loop serial is that probe i sees the table writes of the probes before it -- which only matters
// where probes of one step share a bucket.  Those are found exactly: every probing lane writes its LANE NUMBER to its
// bucket and reads it back; a lane that reads another number shares its bucket, and one ballot per such bucket
// (lanes with that hash) gives the bucket's lanes as a mask.  Inside a bucket the candidate of a lane is the position
// of the next lower lane (its word comes over the lane crossbar, no memory access), the lowest lane keeps the gathered
// table entry.  Then the buckets get back their old entries, and of the lanes up to the winner the highest one of each
// bucket writes its position -- the table ends up exactly as the serial loop leaves it.
// In: ip = first probe position.  Out (true): ip = match position, ref = candidate.  False: ran out of input.
// have_first / first_words: the caller has already loaded the words of the FIRST step (lane l: the dword at ip + first_probe_offset(l),
// 0 where that would leave the input) -- they were requested before the search was known to be needed, so the search starts without a
// round trip to memory.
LZ4HIP_DEVICE int first_probe_offset(int lane) { return lane < 61 ? lane : 2 * lane - 61; }   // skip_sum(67 + lane) - skip_sum(67): 0 .. 60, 61, 63, 65
// One 64-probe step.  ALLV: every lane's probe is known to be inside the input (the caller tested the last lane's), so nothing of the step
// runs under a lane mask -- a guarded load or table access is a saveexec / restore pair and a branch each, a dozen of them per step.
// Returns 0: 64 probes without a match (ip = next probe position), 1: match (ip, ref), 2: ran out of input.
template <bool GENERIC, bool ALLV>
LZ4HIP_DEVICE int wave_find_step(const uint8_t* in, typename FastTable<GENERIC>::entry* table, int& ip, int& ref, int mflimit, int attempts,
                                 bool have_words, uint32_t words)
{
    typedef FastTable<GENERIC> T;
    typedef typename T::entry entry;
    const int lane = wv::lane();
    const uint64_t below_me = (1ull << lane) - 1ull;
    const int a = attempts + lane;
    const int p = ip + skip_sum(a) - skip_sum(attempts);
    const int p_next = p + (a >> 6);
    const bool valid = ALLV || p_next <= mflimit;                    // (monotonic: the probing lanes are a prefix)
    uint32_t w = 0;
    if (have_words) w = valid ? words : 0u;                          // (wave-uniform condition; p + 4 <= n for every valid lane)
    else if (valid) w = load_u32(in + p);
    const uint32_t h = T::hash(w);
    int r = 0;
    if (valid) r = (int)table[h];
    uint32_t rw = 0;                                                 // the candidate's bytes, for lanes alone in their bucket
    if (valid) rw = load_u32(in + r);
    wv::mem_sync();
    if (valid) table[h] = (entry)lane;
    wv::mem_sync();
    uint64_t shared = wv::ballot(valid && (int)table[h] != lane);
    wv::mem_sync();
    uint64_t bucket = 0;                                             // the lanes of this lane's bucket (0: alone)
    while (shared) {                                                 // rare: one round per shared bucket
        const uint32_t hb = wv::readlane(h, wv::ctz64(shared));
        const uint64_t g = wv::ballot(valid && h == hb);
        if (valid && h == hb) bucket = g;
        shared &= ~g;
    }
    const uint64_t lower = bucket & below_me;
    const int prev = lower ? 63 - __builtin_clzll(lower) : lane;
    const uint32_t pw = wv::shuffle(w, prev);
    const int pp = (int)wv::shuffle((uint32_t)p, prev);
    const int cref = lower ? pp : r;
    const bool cand = valid && (!GENERIC || cref >= p - kMaxDistance) && (lower ? pw : rw) == w;   // lz4.c:427 / :654
    const uint64_t stop = wv::ballot(cand || !valid);
    const int m = stop ? wv::ctz64(stop) : 63;                       // the last lane of this step that probes (if valid)
    // table: lanes after the winner restore their bucket, of the others the highest lane of each bucket writes
    if (wv::any(bucket != 0)) {
        if (valid) table[h] = (entry)r;                              // (lanes of one bucket hold the same old entry)
        wv::mem_sync();
        const uint64_t upto_m = m >= 63 ? ~0ull : ((2ull << m) - 1ull);
        const uint64_t higher = bucket & ~below_me & ~(1ull << lane) & upto_m;
        if (valid && lane <= m && higher == 0) table[h] = (entry)p;
    } else if (valid) {
        table[h] = (entry)(lane <= m ? p : r);
    }
    wv::mem_sync();
    if (stop) {
        if (!ALLV && !wv::readlane((uint32_t)valid, m)) return 2;
        ip = (int)wv::readlane((uint32_t)p, m);
        ref = (int)wv::readlane((uint32_t)cref, m);
        return 1;
    }
    ip = (int)wv::readlane((uint32_t)p_next, 63);                    // 64 probes without a match
    return 0;
}

template <bool GENERIC>
LZ4HIP_DEVICE bool wave_find_match(const uint8_t* in, typename FastTable<GENERIC>::entry* table, int& ip, int& ref, int mflimit,
                                   bool have_first = false, uint32_t first_words = 0)
{
    int attempts = 67;
    for (;;) {
        // the LAST lane's probe inside the input (wave-uniform arithmetic): then every lane's is
        const int a63 = attempts + 63;
        const bool all_valid = ip + skip_sum(a63) - skip_sum(attempts) + (a63 >> 6) <= mflimit;
        const bool hw = have_first && attempts == 67;
        const int st = all_valid ? wave_find_step<GENERIC, true>(in, table, ip, ref, mflimit, attempts, hw, first_words)
                                 : wave_find_step<GENERIC, false>(in, table, ip, ref, mflimit, attempts, hw, first_words);
        if (st == 1) return true;
        if (st == 2) return false;
        attempts += 64;
        LZ4HIP_ENC_COUNT(4, 1);
    }
}

template <bool GENERIC>
LZ4HIP_DEVICE int encode_fast_block(const uint8_t* in, int n, uint8_t* out, int cap, unsigned char* table_bytes, bool may_defer = false)
{
    typedef FastTable<GENERIC> T;
    typedef typename T::entry entry;
    entry* table = (entry*)table_bytes;
    const int lane = wv::lane();
    const int mflimit = n - kMfLimit, matchlimit = n - kLastLiterals;
    int ip = 0, anchor = 0, op = 0;
    int sequences = 0, checked_at = 0;                                // (hand-over rule, see kDeferredResult)

    if (n >= kMinLength) {                                            // lz4.c:615
        // fresh zeroed table per block (lz4.c:583 / `new ushort[8192]`, Unsafe.cs:283)
        for (int k = lane * 16; k < kFastTableBytes; k += 64 * 16) {
            Vec16 z = { { 0, 0, 0, 0 } };
            *(Vec16*)(table_bytes + k) = z;
        }
        wv::mem_sync();

        if (GENERIC) {                                                // lz4.c:403: position 0 is inserted
            const uint32_t h0 = T::hash(input_word(in, 0));           // (a no-op on a zeroed table; kept for clarity)
            if (lane == 0) table[h0] = 0;
        }
        ip = 1;                                                       // lz4.c:631: position 0 is never probed
        for (;;) {
            // ---- find a match: lz4.c:642-654, 64 probes of the skip schedule per step ----
            int ref = 0;
            uint32_t cur_word;
            if (!wave_find_match<GENERIC>(in, table, ip, ref, mflimit)) break;

            // ---- catch up: lz4.c:657 ----
            // (the first 256 bytes of the forward count are requested before it: the bytes a catch-up adds in front lie
            //  inside what is already known to be equal, so the forward count from the probe position is the same count)
            uint32_t fwd_round = common_length_round(in, ip + kMinMatch, ref + kMinMatch, matchlimit, 0);
            int caught_up = 0;
            {
                const int room = ip - anchor, bound = room < ref ? room : ref;
                const int back = bound > 0 ? wave_catch_up(in, ip, ref, bound) : 0;
                ip -= back; ref -= back; caught_up = back;
            }

            // ---- literals: lz4.c:660-691 ----
            int ll = ip - anchor;
            int token_at = op++;
            if (op + ll + (ll >> 8) > cap - 8) return 0;             // lz4.c:663
            uint32_t token = ll >= 15 ? 0xF0u : (uint32_t)(ll << 4);
            // (the reference's limit tests under-estimate the length bytes by up to length/65280; it
            //  can only then run past `cap` and return 0 from a later test -- same 0, no stray write)
            if (ll >= 15 && op + (ll - 15) / 255 + 1 + ll > cap) return 0;
            if (ll >= 15) op += put_length_bytes(out + op, ll - 15);
            wave_copy(out + op, in + anchor, ll);
            op += ll;

            for (;;) {
                // ---- offset, match length: lz4.c:693-733 ----
                const uint32_t off = (uint32_t)(ip - ref) & 0xFFFFu;
                if (op + 2 > cap) return 0;                           // (see note above)
                if (lane == 0) { out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8); }
                op += 2;
                ip += kMinMatch; ref += kMinMatch; anchor = ip;
                // (first sequence of a search: counted from the probe position, `caught_up` bytes further on)
                ip += caught_up + wave_common_length(in, ip + caught_up, ref + caught_up, matchlimit, fwd_round);
                caught_up = 0;
                const int extra = ip - anchor;
                if (op + (extra >> 8) > cap - 6) return 0;           // lz4.c:728
                if (extra >= 15 && op + (extra - 15) / 255 + 1 > cap) return 0;   // (see note above)
                token |= extra >= 15 ? 15u : (uint32_t)extra;
                if (lane == 0) out[token_at] = (uint8_t)token;
                if (extra >= 15) op += put_length_bytes(out + op, extra - 15);

                // A block made of short sequences is a chain of dependent steps that 64 lanes cannot shorten; the
                // lane-per-block launch runs 64 such chains per wavefront.  Nothing of this block is final yet.
                if (may_defer && (++sequences % kDeferCheckSequences) == 0) {
                    if (ip - checked_at < kDeferCheckSequences * kDeferBytesPerSequence) return kDeferredResult;
                    checked_at = ip;
                }
                if (ip > mflimit) { anchor = ip; goto tail; }        // lz4.c:736
                // ---- re-seed the table and test the next position: lz4.c:739-751 ----
                {
                    const uint64_t w8 = wv::uniform(load_u64(in + ip - 2));   // bytes ip-2 .. ip+5 (ip <= mflimit): both words in one trip
                    const uint32_t h2 = T::hash((uint32_t)w8);
                    if (lane == 0) table[h2] = (entry)(ip - 2);
                    wv::mem_sync();
                    cur_word = (uint32_t)(w8 >> 16);
                    const uint32_t h = T::hash(cur_word);
                    ref = (int)wv::uniform((uint32_t)table[h]);
                    if (lane == 0) table[h] = (entry)ip;
                }
                const bool in_range = !GENERIC || ref > ip - (kMaxDistance + 1);   // lz4.c:538
                if (!(in_range && input_word(in, ref) == cur_word)) break;
                token_at = op++;                                      // zero-literal sequence (lz4.c:751)
                token = 0;
                fwd_round = common_length_round(in, ip + kMinMatch, ref + kMinMatch, matchlimit, 0);
            }
            anchor = ip++;                                            // lz4.c:754-755
        }
    }
tail:
    {   // ---- last literals: lz4.c:758-767 ----
        const int run = n - anchor;
        if (op + run + 1 + (run - 15 + 255) / 255 > cap) return 0;   // lz4.c:762
        if (lane == 0) out[op] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
        op++;
        if (run >= 15) op += put_length_bytes(out + op, run - 15);
        wave_copy(out + op, in + anchor, run);
        op += run;
    }
    return op;
}

// ---- LZ4_compress64kCtx, second version (round 6): the sequence-dense path --------------------------------------------
// encode_fast_block<false> above spends 238 wave-instructions on a sequence of fuzzer-style data (150 of them scalar;
// profiles/r05/pmc_wave_kernels_512_blocks.json), and a lone wavefront issues one instruction per ~10 cycles: a block's latency IS its
// instruction count.  Two thirds of such data's sequences are the reference's "test next position" case (lz4.c:739-751: zero literals, the
// re-probe at ip hits), so that case gets a straight-line path of its own:
//   * the 8 bytes at ip - 2 (both hash inputs) come out of a REGISTER WINDOW of the input -- lane l holds the dword at wbase + 4 l, three
//     v_readlane_b32 and two scalar shifts give the unaligned qword -- instead of a global load and its round trip;
//   * the candidate test A32(ref) == A32(ip) and the count of the match are ONE operation: 256 bytes of both sides (4 per lane, the ip
//     side requested before the table is touched), xor, ballot, first differing lane: lane 0 differs = no match, else the length;
//   * the whole sequence (token, <= 14 literals, offset) leaves in ONE predicated byte store, lane j writing byte j, once its lengths are
//     known -- the reference writes the token first and ors the match length in later; the bytes are the same.
// The search (wave_find_match, 64 probes per step), the catch-up and the rare shapes (long literal runs, length bytes, the end of the
// block, limited output) are those of the first version, statement for statement.  Same bytes, same return values.
struct FastSeq { int op; bool ok; };
// One sequence: token + literal length bytes + ll literals from in[anchor..] + offset + match length bytes; mlen = the whole match (>= 4).
// The limit tests are those of lz4.c:663 / :728 in the reference's order (and the two of the first version that stand for its stray writes).
// have_lits / lits: lane j >= 1 already holds in[anchor + j - 1] (requested before the search), so the one-store form needs no load.
LZ4HIP_DEVICE FastSeq emit_fast_sequence(const uint8_t* in, uint8_t* out, int cap, int op, int anchor, int ll, uint32_t off, int mlen,
                                         bool have_lits = false, uint32_t lits = 0)
{
    const int lane = wv::lane();
    const int extra = mlen - kMinMatch;
    if (ll < 15 && extra < 15) {
        // token, literals, offset: ll + 3 <= 17 bytes, lane j writes byte j.  (lz4.c:663: op + 1 + ll + (ll >> 8) > cap - 8 fails; what
        // :728 tests afterwards, op + 3 + ll > cap - 6, is the same inequality for these lengths.)
        if (op + 1 + ll > cap - 8) return FastSeq{ 0, false };
        const uint32_t tok = ((uint32_t)ll << 4) | (uint32_t)extra;
        if (lane < ll + 3) {
            uint32_t v = tok;
            if (lane >= 1 && lane <= ll) v = have_lits ? lits : (uint32_t)in[anchor + lane - 1];
            if (lane > ll) v = lane == ll + 1 ? (off & 255u) : (off >> 8);
            out[op + lane] = (uint8_t)v;
        }
        return FastSeq{ op + ll + 3, true };
    }
    int token_at = op++;
    if (op + ll + (ll >> 8) > cap - 8) return FastSeq{ 0, false };                  // lz4.c:663
    if (ll >= 15 && op + (ll - 15) / 255 + 1 + ll > cap) return FastSeq{ 0, false };
    if (ll >= 15) op += put_length_bytes(out + op, ll - 15);
    wave_copy(out + op, in + anchor, ll);
    op += ll;
    if (op + 2 > cap) return FastSeq{ 0, false };
    if (lane == 0) { out[op] = (uint8_t)off; out[op + 1] = (uint8_t)(off >> 8); }
    op += 2;
    if (op + (extra >> 8) > cap - 6) return FastSeq{ 0, false };                    // lz4.c:728
    if (extra >= 15 && op + (extra - 15) / 255 + 1 > cap) return FastSeq{ 0, false };
    if (lane == 0) out[token_at] = (uint8_t)((ll >= 15 ? 0xF0u : (uint32_t)(ll << 4)) | (extra >= 15 ? 15u : (uint32_t)extra));
    if (extra >= 15) op += put_length_bytes(out + op, extra - 15);
    return FastSeq{ op, true };
}

LZ4HIP_DEVICE int encode_fast_block64k(const uint8_t* in, int n, uint8_t* out, int cap, unsigned char* table_bytes, bool may_defer = false)
{
    typedef FastTable<false> T;
    uint16_t* const table = (uint16_t*)table_bytes;
    const int lane = wv::lane();
    const int mflimit = n - kMfLimit, matchlimit = n - kLastLiterals;
    int ip = 0, anchor = 0, op = 0;
    int sequences = 0, checked_at = 0;                                // (hand-over rule, see kDeferredResult)
    LZ4HIP_ENC_DECL();

    if (n >= kMinLength) {                                            // lz4.c:615
        for (int k = lane * 16; k < kFastTableBytes; k += 64 * 16) {  // fresh zeroed table per block (lz4.c:583)
            Vec16 z = { { 0, 0, 0, 0 } };
            *(Vec16*)(table_bytes + k) = z;
        }
        wv::mem_sync();
        // the register window: lane l holds in[wbase + 4 l .. + 4) (0 past the end); a qword at q comes out of it while 0 <= q - wbase <= 244
        uint32_t win = 0;
        int wbase = -(1 << 20);
        // the words of the next search's first 64-probe step, requested by the test-next-position iteration before it knows whether it hits
        bool have_first = false;
        uint32_t first_words = 0;
        const int first_off = first_probe_offset(lane);
        ip = 1;                                                       // lz4.c:631: position 0 is never probed
        for (;;) {
            // ---- find a match: lz4.c:642-654, 64 probes of the skip schedule per step ----
            int ref = 0;
            LZ4HIP_ENC_T0();
            // (the literals of the sequence this search ends in start at `anchor`: lane j >= 1 fetches in[anchor + j - 1] now -- it is there
            //  long before the sequence is emitted)
            uint32_t pre_lits = 0;
            if (lane >= 1 && anchor + lane - 1 < n) pre_lits = in[anchor + lane - 1];
            if (!wave_find_match<false>(in, table, ip, ref, mflimit, have_first, first_words)) break;
            have_first = false;
            LZ4HIP_ENC_T(0);
            // ---- catch up (lz4.c:657), literals, offset, match length (lz4.c:660-733) ----
            // Away from the end of the block all four loads -- 64 bytes before both positions for the catch-up, 256 bytes behind both for the
            // count -- are issued together, unconditionally, and waited for ONCE (written as a guarded round each, the compiler waits for the
            // count's loads before it even issues the catch-up's: two round trips).  The catch-up bytes are clamped to the block's start and
            // only compared below `bound`; the count is from the probe position (what a catch-up adds in front lies inside what is known to be equal).
            {
                int mlen;
                const int room = ip - anchor, bound = room < ref ? room : ref;
                if (ip + kMinMatch + 256 <= matchlimit) {
                    const int ia = ip - 1 - lane, ib = ref - 1 - lane;
                    const uint32_t ca = in[ia > 0 ? ia : 0], cb = in[ib > 0 ? ib : 0];
                    const uint32_t diff = load_u32(in + ip + kMinMatch + 4 * lane) ^ load_u32(in + ref + kMinMatch + 4 * lane);
                    const uint64_t differs = wv::ballot(lane >= bound || ca != cb);
                    int back = differs ? wv::ctz64(differs) : 64;
                    if (back == 64 && bound > 64) back = 64 + wave_catch_up(in, ip - 64, ref - 64, bound - 64);
                    const uint64_t stop = wv::ballot(diff != 0);
                    int fwd;
                    if (stop) { const int first = wv::ctz64(stop); fwd = first * 4 + (__builtin_ctz(wv::readlane(diff, first)) >> 3); }
                    else fwd = 256 + wave_common_length(in, ip + kMinMatch + 256, ref + kMinMatch + 256, matchlimit);
                    ip -= back; ref -= back;
                    mlen = kMinMatch + back + fwd;
                } else {
                    const uint32_t fwd_round = common_length_round(in, ip + kMinMatch, ref + kMinMatch, matchlimit, 0);
                    const int back = bound > 0 ? wave_catch_up(in, ip, ref, bound) : 0;
                    ip -= back; ref -= back;
                    mlen = kMinMatch + back + wave_common_length(in, ip + kMinMatch + back, ref + kMinMatch + back, matchlimit, fwd_round);
                }
                const FastSeq e = emit_fast_sequence(in, out, cap, op, anchor, ip - anchor, (uint32_t)(ip - ref) & 0xFFFFu, mlen, true, pre_lits);
                if (!e.ok) return 0;
                op = e.op;
                ip += mlen; anchor = ip;
            }
            LZ4HIP_ENC_T(1);
            // ---- test next position: lz4.c:736-751, as long as it hits ----
            for (;;) {
                LZ4HIP_ENC_T0();
                if (may_defer && (++sequences % kDeferCheckSequences) == 0) {
                    if (ip - checked_at < kDeferCheckSequences * kDeferBytesPerSequence) return kDeferredResult;
                    checked_at = ip;
                }
                if (ip > mflimit) { anchor = ip; goto tail; }        // lz4.c:736
                // bytes ip-2 .. ip+5 (ip <= mflimit: inside the block) out of the window; re-based here -- BEFORE this iteration's loads are
                // requested, and waited for inside the branch: a wait behind the join would make every iteration wait for those loads too
                const int q = ip - 2;
                int rel = q - wbase;
                if ((unsigned)rel > 244u) {
                    wbase = q; rel = 0;
                    const int pos = q + 4 * lane;
                    win = pos + 4 <= n ? load_u32(in + pos) : 0u;
                    wv::wait_vector_memory();
                    LZ4HIP_KEEP(win);
                }
                const bool wide = ip + 256 <= matchlimit;             // 256 bytes from ip may be read and counted (wave-uniform)
                // Both loads are issued in EVERY iteration, from addresses clamped to the block (n >= 13 here): a load under a condition
                // needs its register zeroed first, and that write has to wait for whatever load into the same register is still in
                // flight -- on gfx9 together with every store issued before it.  What a clamped lane reads is never used (a4 only when
                // `wide`, first_words only by lanes whose probe position is valid).
                const int last4 = n - 4;
                const int pa = ip + 4 * lane, pf = ip + 1 + first_off;
                const uint32_t a4 = load_u32(in + (pa < last4 ? pa : last4));          // the ip side of test + count, on its way while the table is looked up
                first_words = load_u32(in + (pf < last4 ? pf : last4));               // if this test misses, the search starts at ip + 1: its first step's words
                have_first = true;
                const int k = rel >> 2;
                const uint32_t sh = ((uint32_t)rel & 3u) * 8u;
                const uint32_t d0 = wv::readlane(win, k), d1 = wv::readlane(win, k + 1), d2 = wv::readlane(win, k + 2);
                const uint32_t w_m2 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);     // bytes ip-2 .. ip+1
                const uint32_t w_p2 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);     // bytes ip+2 .. ip+5
                const uint32_t cur_word = (w_m2 >> 16) | (w_p2 << 16);                   // bytes ip .. ip+3
                const uint32_t h2 = T::hash(w_m2), h = T::hash(cur_word);
                LZ4HIP_ENC_T(4);
                table[h2] = (uint16_t)(ip - 2);                       // (every lane stores the same value: no lane mask to set up)
                wv::mem_sync();
                const int ref2 = (int)wv::uniform((uint32_t)table[h]);
                table[h] = (uint16_t)ip;
                LZ4HIP_ENC_T(5);
                int mlen;
                if (wide) {
                    const uint32_t diff = a4 ^ load_u32(in + ref2 + 4 * lane);
                    const uint64_t stop = wv::ballot(diff != 0);
                    if (stop & 1ull) { LZ4HIP_ENC_T(3); break; }      // A32(ref) != A32(ip)
                    if (stop) {
                        const int first = wv::ctz64(stop);
                        mlen = first * 4 + (__builtin_ctz(wv::readlane(diff, first)) >> 3);
                    } else {
                        mlen = 256 + wave_common_length(in, ip + 256, ref2 + 256, matchlimit);
                    }
                } else {
                    if (input_word(in, ref2) != cur_word) { LZ4HIP_ENC_T(3); break; }
                    mlen = kMinMatch + wave_common_length(in, ip + kMinMatch, ref2 + kMinMatch, matchlimit);
                }
                LZ4HIP_ENC_T(6);
                const FastSeq e = emit_fast_sequence(in, out, cap, op, ip, 0, (uint32_t)(ip - ref2) & 0xFFFFu, mlen);   // zero-literal sequence (lz4.c:751)
                if (!e.ok) return 0;
                op = e.op;
                ip += mlen; anchor = ip;
                LZ4HIP_ENC_T(2);
            }
            anchor = ip++;                                            // lz4.c:754-755
        }
    }
tail:
    {   // ---- last literals: lz4.c:758-767 ----
        const int run = n - anchor;
        if (op + run + 1 + (run - 15 + 255) / 255 > cap) return 0;   // lz4.c:762
        if (lane == 0) out[op] = (uint8_t)(run >= 15 ? 0xF0 : (run << 4));
        op++;
        if (run >= 15) op += put_length_bytes(out + op, run - 15);
        wave_copy(out + op, in + anchor, run);
        op += run;
    }
    LZ4HIP_ENC_FLUSH();
    return op;
}

// One wavefront (= one workgroup of 64 threads) per block; 16 KiB of dynamic LDS per workgroup,
// so up to 10 blocks are resident per CU.
// flags: kEncodeOnlyGeneric: handle just the blocks of LZ4_64KLIMIT bytes and more; kEncodeMayDefer: blocks below
// LZ4_64KLIMIT made of short sequences get kDeferredResult instead of being finished (a second launch takes them).
enum EncodeKernelFlags { kEncodeOnlyGeneric = 1, kEncodeMayDefer = 2 };
// V64K: which version of the 64k encoder runs (2 = encode_fast_block64k, the product; 1 = the first version, instantiated in tuning builds only: A/B runs)
// WAVES: wavefronts (= blocks) per workgroup, each with its own 16 KiB of the workgroup's dynamic LDS.  gfx950 allocates LDS in granules of 1 280 bytes
// (128 per CU): one table = 13 granules = nine one-block workgroups per CU, five tables = 64 granules exactly = two workgroups = TEN blocks per CU.
// The launch site picks 5 wherever that saves a residency round (lz4hip_api.hip: encoder_five_blocks_per_workgroup); no barrier anywhere: the
// wavefronts of a workgroup share nothing but the allocation.
template <int V64K = 2, int WAVES = 1>
__global__ void __launch_bounds__(64 * WAVES) encode_fast_kernel(Batch b, int flags)
{
    const int only_generic = flags & kEncodeOnlyGeneric;
    LZ4HIP_DYN_LDS(lds_all);
    unsigned char* const lds = lds_all + (WAVES > 1 ? (size_t)wv::wave_in_block() * kFastTableBytes : 0);
    const int64_t blk = WAVES > 1 ? (int64_t)blockIdx.x * WAVES + wv::wave_in_block() : (int64_t)blockIdx.x;
    if (blk >= b.n_blocks) return;
    const int n = wv::uniform(batch_src_len(b, blk));
    if (only_generic && n < k64kLimit) return;
    const int cap = wv::uniform(batch_dst_cap(b, blk));
    const uint8_t* src = batch_src(b, blk);
    uint8_t* dst = batch_dst(b, blk);
    int r;
    if (n < k64kLimit) r = V64K == 2 ? encode_fast_block64k(src, n, dst, cap, lds, (flags & kEncodeMayDefer) != 0)          // lz4.c:783-785
                                     : encode_fast_block<false>(src, n, dst, cap, lds, (flags & kEncodeMayDefer) != 0);
    else               r = encode_fast_block<true>(src, n, dst, cap, lds);
    if (wv::lane() == 0) b.result[blk] = r;
}

}  // namespace lz4hip
