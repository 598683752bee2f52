/*
 * lz4_oracle.c -- TEST INFRASTRUCTURE ONLY (see lz4_oracle.h).
 *
 * From-scratch CPU restatement of the reference's LZ4 block codec.  Written against positions
 * (ints) instead of pointers, with exact-length copies instead of the reference's 8-byte "wild"
 * copies: only the FINAL bytes and the return codes are part of the contract.
 * Every routine cites the reference lines it follows; tests/test_oracle_vs_ref.py pins each one
 * against the reference's own C compiled in place (oracle/_ref).
 */
#include "lz4_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- format constants: original/lz4.c:182-203,566-570; original/lz4hc.c:173-195 ---------- */
enum {
    MIN_MATCH      = 4,
    LAST_LITERALS  = 5,      /* the last 5 bytes of a block are always literals                 */
    MF_LIMIT       = 12,     /* no match may start within the last 12 bytes                     */
    MIN_LENGTH     = 13,     /* shorter inputs are emitted as one literal run                   */
    MAX_DIST       = 65535,
    LIMIT_64K      = 65536 + 11,   /* LZ4_64KLIMIT: inputs below this use the u16[8192] table   */
    HC_ATTEMPTS    = 256,
    HC_OPTIMAL_ML  = 18
};
#define GOLDEN 2654435761u

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash13(const uint8_t* p) { return (rd32(p) * GOLDEN) >> 19; } /* lz4.c:569 */
static inline uint32_t hash12(const uint8_t* p) { return (rd32(p) * GOLDEN) >> 20; } /* lz4.c:248 */
static inline uint32_t hash15(const uint8_t* p) { return (rd32(p) * GOLDEN) >> 17; } /* lz4hc.c:245 */

int lz4o_compress_bound(int isize) { return isize + isize / 255 + 16; }   /* lz4.h:85-86 */

/* Exact common-prefix length of in[a..] and in[b..] with a < limit as the bound on `a`.
 * The reference counts 8 bytes at a time with XOR + ctz and 4/2/1-byte tails
 * (lz4.c:698-721, lz4hc.c:376-391); that procedure yields exactly this number. */
static inline int common_len(const uint8_t* in, int a, int b, int limit)
{
    int n = 0;
    while (a + n < limit && in[a + n] == in[b + n]) n++;
    return n;
}

/* ---- sequence emission shared by both fast variants ------------------------------------- */
typedef struct { uint8_t* out; int op; int cap; } sink_t;

/* literal-length part of a sequence: lz4.c:660-691.  Returns 0 on output-limit hit. */
static int put_literals(sink_t* s, const uint8_t* in, int anchor, int litlen, int* token_at)
{
    *token_at = s->op++;
    if (s->op + litlen + (litlen >> 8) > s->cap - 8) return 0;          /* lz4.c:663 */
    if (litlen >= 15) {
        int rest = litlen - 15;
        s->out[*token_at] = 0xF0;
        while (rest > 254) { s->out[s->op++] = 255; rest -= 255; }
        s->out[s->op++] = (uint8_t)rest;
    } else {
        s->out[*token_at] = (uint8_t)(litlen << 4);
    }
    memcpy(s->out + s->op, in + anchor, (size_t)litlen);
    s->op += litlen;
    return 1;
}

/* match-length part: lz4.c:724-733.  `extra` = match length - 4. */
static int put_matchlen(sink_t* s, int token_at, int extra)
{
    if (s->op + (extra >> 8) > s->cap - 6) return 0;                    /* lz4.c:728 */
    if (extra >= 15) {
        s->out[token_at] += 15;
        extra -= 15;
        while (extra > 254) { s->out[s->op++] = 255; extra -= 255; }   /* same bytes as the 510-pair loop */
        s->out[s->op++] = (uint8_t)extra;
    } else {
        s->out[token_at] += (uint8_t)extra;
    }
    return 1;
}

/* trailing literal run: lz4.c:758-767 */
static int put_last_literals(sink_t* s, const uint8_t* in, int anchor, int n)
{
    int run = n - anchor;
    if (s->op + run + 1 + (run - 15 + 255) / 255 > s->cap) return 0;    /* lz4.c:762 */
    if (run >= 15) {
        int rest = run - 15;
        s->out[s->op++] = 0xF0;
        while (rest > 254) { s->out[s->op++] = 255; rest -= 255; }
        s->out[s->op++] = (uint8_t)rest;
    } else {
        s->out[s->op++] = (uint8_t)(run << 4);
    }
    memcpy(s->out + s->op, in + anchor, (size_t)run);
    s->op += run;
    return 1;
}

/* ---- fast encoder, 64k variant: original/lz4.c:573-771 ---------------------------------- */
static int fast_64k(const uint8_t* in, int n, uint8_t* out, int cap)
{
    uint16_t table[8192];
    sink_t s = { out, 0, cap };
    int ip = 0, anchor = 0;
    const int mflimit = n - MF_LIMIT, matchlimit = n - LAST_LITERALS;

    if (n < MIN_LENGTH) goto tail;                                       /* lz4.c:615 */
    memset(table, 0, sizeof table);                                     /* fresh table per call */

    ip = 1;                                                              /* position 0 is never inserted, lz4.c:631 */
    uint32_t fwd_hash = hash13(in + ip);
    for (;;) {
        int attempts = 67;                                               /* (1<<6)+3, lz4.c:636 */
        int probe = ip, ref, token_at;
        /* search: read-then-overwrite the bucket with the hash computed one step earlier */
        for (;;) {
            uint32_t h = fwd_hash;
            int step = attempts++ >> 6;
            ip = probe;
            probe = ip + step;
            if (probe > mflimit) goto tail;                              /* lz4.c:648 */
            fwd_hash = hash13(in + probe);
            ref = table[h];                                              /* empty bucket == position 0 */
            table[h] = (uint16_t)ip;
            if (rd32(in + ref) == rd32(in + ip)) break;                  /* no distance check here */
        }
        /* catch-up: lz4.c:657 */
        while (ip > anchor && ref > 0 && in[ip - 1] == in[ref - 1]) { ip--; ref--; }

        if (!put_literals(&s, in, anchor, ip - anchor, &token_at)) return 0;
        for (;;) {
            /* offset + match length: lz4.c:693-733 */
            uint32_t off = (uint32_t)(ip - ref) & 0xFFFF;
            out[s.op++] = (uint8_t)off; out[s.op++] = (uint8_t)(off >> 8);
            ip += MIN_MATCH; ref += MIN_MATCH; anchor = ip;
            ip += common_len(in, ip, ref, matchlimit);
            if (!put_matchlen(&s, token_at, ip - anchor)) return 0;

            if (ip > mflimit) { anchor = ip; goto tail; }                /* lz4.c:736 */
            table[hash13(in + ip - 2)] = (uint16_t)(ip - 2);             /* lz4.c:739 */
            uint32_t h = hash13(in + ip);                                /* lz4.c:742-749 */
            ref = table[h];
            table[h] = (uint16_t)ip;
            if (rd32(in + ref) != rd32(in + ip)) break;
            token_at = s.op++;                                           /* zero-literal sequence, lz4.c:751 */
            out[token_at] = 0;
        }
        anchor = ip++;                                                   /* lz4.c:754-755 */
        fwd_hash = hash13(in + ip);
    }
tail:
    if (!put_last_literals(&s, in, anchor, n)) return 0;
    return s.op;
}

/* ---- fast encoder, generic variant (inputs >= 65547 bytes): original/lz4.c:345-562 ------ */
static int fast_generic(const uint8_t* in, int n, uint8_t* out, int cap)
{
    uint32_t* table = (uint32_t*)calloc(4096, sizeof(uint32_t));
    sink_t s = { out, 0, cap };
    int ip = 0, anchor = 0, result = 0;
    const int mflimit = n - MF_LIMIT, matchlimit = n - LAST_LITERALS;
    if (!table) return 0;
    if (n < MIN_LENGTH) goto tail;

    table[hash12(in)] = 0;                                               /* position 0 IS inserted, lz4.c:403 */
    ip = 1;
    uint32_t fwd_hash = hash12(in + ip);
    for (;;) {
        int attempts = 67;
        int probe = ip, ref, token_at;
        for (;;) {
            uint32_t h = fwd_hash;
            int step = attempts++ >> 6;
            ip = probe;
            probe = ip + step;
            if (probe > mflimit) goto tail;
            fwd_hash = hash12(in + probe);
            ref = (int)table[h];
            table[h] = (uint32_t)ip;
            if (ref >= ip - MAX_DIST && rd32(in + ref) == rd32(in + ip)) break;   /* lz4.c:427 */
        }
        while (ip > anchor && ref > 0 && in[ip - 1] == in[ref - 1]) { ip--; ref--; }

        if (!put_literals(&s, in, anchor, ip - anchor, &token_at)) goto fail;
        for (;;) {
            uint32_t off = (uint32_t)(ip - ref) & 0xFFFF;
            out[s.op++] = (uint8_t)off; out[s.op++] = (uint8_t)(off >> 8);
            ip += MIN_MATCH; ref += MIN_MATCH; anchor = ip;
            ip += common_len(in, ip, ref, matchlimit);
            if (!put_matchlen(&s, token_at, ip - anchor)) goto fail;

            if (ip > mflimit) { anchor = ip; goto tail; }
            table[hash12(in + ip - 2)] = (uint32_t)(ip - 2);
            uint32_t h = hash12(in + ip);
            ref = (int)table[h];
            table[h] = (uint32_t)ip;
            if (!(ref > ip - (MAX_DIST + 1) && rd32(in + ref) == rd32(in + ip))) break;   /* lz4.c:538 */
            token_at = s.op++;
            out[token_at] = 0;
        }
        anchor = ip++;
        fwd_hash = hash12(in + ip);
    }
tail:
    if (put_last_literals(&s, in, anchor, n)) result = s.op;
fail:
    free(table);
    return result;
}

int lz4o_compress_limited(const uint8_t* src, uint8_t* dst, int isize, int max_out)
{
    /* lz4.c:783-785 */
    return isize < LIMIT_64K ? fast_64k(src, isize, dst, max_out) : fast_generic(src, isize, dst, max_out);
}

/* ---- decoders ----------------------------------------------------------------------------- */
/* Byte-wise overlapped copy out[op+i] = out[ref+i]: the semantics the reference implements with
 * dec32table/dec64table pointer fix-ups + wild copies (lz4.c:832-835,869-905). */
static inline void copy_match(uint8_t* out, int op, int ref, int len)
{
    for (int i = 0; i < len; i++) out[op + i] = out[ref + i];
}

/* original/lz4.c:812-914 */
int lz4o_uncompress(const uint8_t* src, uint8_t* dst, int osize)
{
    long ip = 0, op = 0;
    const long oend = osize;
    for (;;) {
        unsigned token = src[ip++];
        long len = token >> 4;
        if (len == 15) { unsigned b; do { b = src[ip++]; len += b; } while (b == 255); }   /* lz4.c:844 */

        long lit_end = op + len;
        if (lit_end > oend - 8) {                                        /* lz4.c:851-858 */
            if (lit_end != oend) return (int)-ip;
            memcpy(dst + op, src + ip, (size_t)len);
            ip += len;
            return (int)ip;                                              /* normal end: bytes consumed */
        }
        memcpy(dst + op, src + ip, (size_t)len);
        ip += len; op = lit_end;

        long ref = op - (src[ip] | (src[ip + 1] << 8));                  /* lz4.c:862 */
        ip += 2;
        if (ref < 0) return (int)-ip;                                    /* lz4.c:863 */

        len = token & 15;
        if (len == 15) { while (src[ip] == 255) { len += 255; ip++; } len += src[ip++]; }  /* lz4.c:866 */
        len += MIN_MATCH;

        if (op + len > oend - 12 && op + len > oend - LAST_LITERALS) return (int)-ip;      /* lz4.c:887-893 */
        copy_match(dst, (int)op, (int)ref, (int)len);
        op += len;
    }
}

/* original/lz4.c:916-1044 */
int lz4o_uncompress_unknown(const uint8_t* src, uint8_t* dst, int isize, int max_out)
{
    long ip = 0, op = 0;
    const long iend = isize, oend = max_out;
    if (ip == iend) return 0;                                            /* lz4.c:946: -(0) */
    for (;;) {
        unsigned token = src[ip++];
        long len = token >> 4;
        if (len == 15) {                                                 /* lz4.c:957-961 */
            unsigned b = 255;
            while (ip < iend && b == 255) { b = src[ip++]; len += b; }
        }
        long lit_end = op + len;
        if (lit_end > oend - MF_LIMIT || ip + len > iend - 8) {          /* lz4.c:965-975 */
            if (lit_end > oend) return (int)-ip;
            if (ip + len != iend) return (int)-ip;
            memcpy(dst + op, src + ip, (size_t)len);
            op += len;
            return (int)op;                                              /* bytes produced */
        }
        memcpy(dst + op, src + ip, (size_t)len);
        ip += len; op = lit_end;

        long ref = op - (src[ip] | (src[ip + 1] << 8));
        ip += 2;
        if (ref < 0) return (int)-ip;                                    /* lz4.c:980 */

        len = token & 15;
        if (len == 15) {                                                 /* lz4.c:983-997 */
            while (ip < iend - (LAST_LITERALS + 1)) {
                unsigned b = src[ip++];
                len += b;
                if (b != 255) break;
            }
        }
        len += MIN_MATCH;
        if (op + len > oend - 12 && op + len > oend - LAST_LITERALS) return (int)-ip;      /* lz4.c:1018-1024 */
        copy_match(dst, (int)op, (int)ref, (int)len);
        op += len;
    }
}

/* ---- LZ4HC ---------------------------------------------------------------------------------- */
/* state: original/lz4hc.c:231-237,330-337 (LZ4_ARCH64=1: offset-typed heads, nextToUpdate = 1) */
typedef struct {
    uint32_t head[32768];      /* zero-filled: empty bucket == position 0 */
    uint16_t chain[65536];     /* 0xFFFF-filled; slot = position & 0xFFFF  */
    long     next;             /* first position not yet inserted          */
    const uint8_t* in;
} hc_state;

/* lz4hc.c:358-373 */
static void hc_insert_upto(hc_state* st, long ip)
{
    while (st->next < ip) {
        long p = st->next;
        uint32_t h = hash15(st->in + p);
        uint64_t delta = (uint64_t)(p - (long)st->head[h]);
        if (delta > MAX_DIST) delta = MAX_DIST;
        st->chain[p & 0xFFFF] = (uint16_t)delta;
        st->head[h] = (uint32_t)p;
        st->next++;
    }
}

/* lz4hc.c:394-459 */
static int hc_best_match(hc_state* st, long ip, long matchlimit, long* match_at)
{
    const uint8_t* in = st->in;
    int attempts = HC_ATTEMPTS;
    long repl = 0, ml = 0;
    uint16_t delta = 0;

    hc_insert_upto(st, ip);
    long ref = st->head[hash15(in + ip)];

    if (ref >= ip - 4) {                                                 /* repeat detection, lz4hc.c:411-421 */
        if (rd32(in + ref) == rd32(in + ip)) {
            delta = (uint16_t)(ip - ref);
            repl = ml = common_len(in, (int)ip + 4, (int)ref + 4, (int)matchlimit) + 4;
            *match_at = ref;
        }
        ref -= st->chain[ref & 0xFFFF];
    }
    while (ref >= ip - MAX_DIST && attempts) {                           /* lz4hc.c:424-434 */
        attempts--;
        if (ref < 0) break;    /* unreachable on the reference's flows (it would read before the buffer) */
        if (in[ref + ml] == in[ip + ml] && rd32(in + ref) == rd32(in + ip)) {
            long cand = common_len(in, (int)ip + 4, (int)ref + 4, (int)matchlimit) + 4;
            if (cand > ml) { ml = cand; *match_at = ref; }
        }
        ref -= st->chain[ref & 0xFFFF];
    }
    if (repl) {                                                          /* pre-fill, lz4hc.c:437-455 */
        long p = ip, end = ip + repl - 3;
        while (p < end - delta) { st->chain[p & 0xFFFF] = delta; p++; }
        do {
            st->chain[p & 0xFFFF] = delta;
            st->head[hash15(in + p)] = (uint32_t)p;
            p++;
        } while (p < end);
        st->next = end;
    }
    return (int)ml;
}

/* lz4hc.c:462-518 */
static int hc_wider_match(hc_state* st, long ip, long start_limit, long matchlimit, int longest,
                          long* match_at, long* start_at)
{
    const uint8_t* in = st->in;
    int attempts = HC_ATTEMPTS;
    long back = ip - start_limit;

    hc_insert_upto(st, ip);
    long ref = st->head[hash15(in + ip)];
    while (ref >= ip - MAX_DIST && attempts) {
        attempts--;
        if (ref < 0) break;    /* see hc_best_match */
        if (in[start_limit + longest] == in[ref - back + longest] && rd32(in + ref) == rd32(in + ip)) {
            long fwd_end = ip + 4 + common_len(in, (int)ip + 4, (int)ref + 4, (int)matchlimit);
            long s = ip, r = ref;
            while (s > start_limit && r > 0 && in[s - 1] == in[r - 1]) { s--; r--; }   /* lz4hc.c:505 */
            if (fwd_end - s > longest) { longest = (int)(fwd_end - s); *match_at = r; *start_at = s; }
        }
        ref -= st->chain[ref & 0xFFFF];
    }
    return longest;
}

/* lz4hc.c:521-550.  Returns 0 on output-limit hit. */
static int hc_emit(sink_t* s, const uint8_t* in, long* ip, long* anchor, int ml, long ref)
{
    int litlen = (int)(*ip - *anchor);
    int token_at = s->op++;
    if (s->op + litlen + 8 + (litlen >> 8) > s->cap) return 0;            /* lz4hc.c:529 */
    if (litlen >= 15) {
        int rest = litlen - 15;
        s->out[token_at] = 0xF0;
        while (rest > 254) { s->out[s->op++] = 255; rest -= 255; }
        s->out[s->op++] = (uint8_t)rest;
    } else {
        s->out[token_at] = (uint8_t)(litlen << 4);
    }
    memcpy(s->out + s->op, in + *anchor, (size_t)litlen);
    s->op += litlen;
    uint32_t off = (uint32_t)(*ip - ref) & 0xFFFF;
    s->out[s->op++] = (uint8_t)off; s->out[s->op++] = (uint8_t)(off >> 8);
    int extra = ml - MIN_MATCH;
    if (s->op + 6 + (litlen >> 8) > s->cap) return 0;                     /* lz4hc.c:541: uses the LITERAL length */
    if (extra >= 15) {
        s->out[token_at] += 15;
        extra -= 15;
        while (extra > 254) { s->out[s->op++] = 255; extra -= 255; }
        s->out[s->op++] = (uint8_t)extra;
    } else {
        s->out[token_at] += (uint8_t)extra;
    }
    *ip += ml;
    *anchor = *ip;
    return 1;
}

/* lz4hc.c:557-742 (lazy 3-match parser), driven through lz4hc.c:745-755 */
int lz4o_compress_hc_limited(const uint8_t* src, uint8_t* dst, int isize, int max_out)
{
    hc_state* st = (hc_state*)malloc(sizeof *st);
    if (!st) return 0;
    memset(st->head, 0, sizeof st->head);
    memset(st->chain, 0xFF, sizeof st->chain);
    st->next = 1;
    st->in = src;

    sink_t s = { dst, 0, max_out };
    const long n = isize, mflimit = n - MF_LIMIT, matchlimit = n - LAST_LITERALS;
    long ip = 0, anchor = 0;
    long ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0, ref0;
    int ml, ml2, ml3, ml0, result = 0;

    ip++;
    while (ip < mflimit) {
        ml = hc_best_match(st, ip, matchlimit, &ref);
        if (!ml) { ip++; continue; }
        start0 = ip; ref0 = ref; ml0 = ml;

        int search3 = 0;              /* 0: at "_Search2", 1: at "_Search3" */
        for (;;) {
            if (!search3) {
                ml2 = (ip + ml < mflimit)
                    ? hc_wider_match(st, ip + ml - 2, ip + 1, matchlimit, ml, &ref2, &start2) : ml;
                if (ml2 == ml) {                                         /* lz4hc.c:599-603 */
                    if (!hc_emit(&s, src, &ip, &anchor, ml, ref)) goto done;
                    break;
                }
                if (start0 < ip && start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }   /* :605-613 */
                if (start2 - ip < 3) { ml = ml2; ip = start2; ref = ref2; continue; }         /* :616-622 */
            }
            search3 = 1;
            if (start2 - ip < HC_OPTIMAL_ML) {                           /* lz4hc.c:628-641 */
                int new_ml = ml > HC_OPTIMAL_ML ? HC_OPTIMAL_ML : ml;
                if (ip + new_ml > start2 + ml2 - MIN_MATCH) new_ml = (int)(start2 - ip) + ml2 - MIN_MATCH;
                int corr = new_ml - (int)(start2 - ip);
                if (corr > 0) { start2 += corr; ref2 += corr; ml2 -= corr; }
            }
            ml3 = (start2 + ml2 < mflimit)
                ? hc_wider_match(st, start2 + ml2 - 3, start2, matchlimit, ml2, &ref3, &start3) : ml2;
            if (ml3 == ml2) {                                            /* two sequences, lz4hc.c:648-657 */
                if (start2 < ip + ml) ml = (int)(start2 - ip);
                if (!hc_emit(&s, src, &ip, &anchor, ml, ref)) goto done;
                ip = start2;
                if (!hc_emit(&s, src, &ip, &anchor, ml2, ref2)) goto done;
                break;
            }
            if (start3 < ip + ml + 3) {                                  /* lz4hc.c:659-691 */
                if (start3 >= ip + ml) {
                    if (start2 < ip + ml) {
                        int corr = (int)(ip + ml - start2);
                        start2 += corr; ref2 += corr; ml2 -= corr;
                        if (ml2 < MIN_MATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                    }
                    if (!hc_emit(&s, src, &ip, &anchor, ml, ref)) goto done;
                    ip = start3; ref = ref3; ml = ml3;
                    start0 = start2; ref0 = ref2; ml0 = ml2;
                    search3 = 0;
                    continue;
                }
                start2 = start3; ref2 = ref3; ml2 = ml3;
                continue;
            }
            if (start2 < ip + ml) {                                      /* lz4hc.c:695-715 */
                if (start2 - ip < 15) {
                    if (ml > HC_OPTIMAL_ML) ml = HC_OPTIMAL_ML;
                    if (ip + ml > start2 + ml2 - MIN_MATCH) ml = (int)(start2 - ip) + ml2 - MIN_MATCH;
                    int corr = ml - (int)(start2 - ip);
                    if (corr > 0) { start2 += corr; ref2 += corr; ml2 -= corr; }
                } else {
                    ml = (int)(start2 - ip);
                }
            }
            if (!hc_emit(&s, src, &ip, &anchor, ml, ref)) goto done;
            ip = start2; ref = ref2; ml = ml2;
            start2 = start3; ref2 = ref3; ml2 = ml3;
        }
    }
    {   /* last literals, lz4hc.c:730-738 */
        int run = (int)(n - anchor);
        if ((long)s.op + run + 1 + (run + 255 - 15) / 255 > (long)(uint32_t)max_out) goto done;
        if (run >= 15) {
            int rest = run - 15;
            dst[s.op++] = 0xF0;
            while (rest > 254) { dst[s.op++] = 255; rest -= 255; }
            dst[s.op++] = (uint8_t)rest;
        } else {
            dst[s.op++] = (uint8_t)(run << 4);
        }
        memcpy(dst + s.op, src + anchor, (size_t)run);
        s.op += run;
        result = s.op;
    }
done:
    free(st);
    return result;
}
