/* synth.c -- TEST INFRASTRUCTURE ONLY (see synth.h). */
#include "synth.h"
#include <string.h>

static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t block_key(uint64_t seed, uint64_t block) { return mix64(mix64(seed) + block); }

/* original/fuzzer.c:81-85 (FUZ_rand) with PRIME1/PRIME2 of :54-55 */
static inline uint32_t lcg(uint32_t* s) { *s = *s * 2654435761u + 2246822519u; return *s; }

static void fill_random(uint64_t key, uint8_t* out, int len)
{
    int i = 0;
    for (uint64_t w = 0; i < len; w++) {
        uint64_t v = mix64(key + w * 0xD1342543DE82EF95ull);
        for (int k = 0; k < 8 && i < len; k++, i++) out[i] = (uint8_t)(v >> (8 * k));
    }
}

/* original/fuzzer.c:149-168, one buffer, seeded per block */
static void fill_fuz(uint64_t key, uint8_t* out, int len)
{
    uint32_t s = (uint32_t)key, seeds[4], cur = 3266489917u;
    lcg(&s);
    for (int j = 0; j < 4; j++) { seeds[j] = lcg(&s) << 8; seeds[j] ^= (lcg(&s) >> 8) & 65535; }
    for (int j = 0; j < len; j++) {
        int32_t k = (int32_t)lcg(&s);
        if (j == 0 || ((k >> 10) % 10) == 0) cur = seeds[(lcg(&s) >> 16) & 3];
        if (((k >> 8) & 255) == 0) {
            uint32_t q = (lcg(&s) >> 16) & 3;
            seeds[q] = lcg(&s) << 8;
            seeds[q] ^= (lcg(&s) >> 8) & 65535;
        }
        out[j] = (uint8_t)(lcg(&cur) >> 16);
    }
}

static void fill_records(uint64_t key, uint8_t* out, int len)
{
    uint32_t s = (uint32_t)(key >> 16) | 1u;
    uint32_t lit_left = 0, mat_left = 0, dist = 0;
    for (int pos = 0; pos < len; pos++) {
        uint8_t b;
        if (lit_left == 0 && mat_left == 0) {
            uint32_t r = lcg(&s);
            lit_left = 4 + ((r >> 8) % 24);
            mat_left = pos > 0 ? 8 + ((r >> 16) % 88) : 0;
        }
        if (lit_left) {
            b = (uint8_t)(0x20 + ((lcg(&s) >> 16) & 63));
            lit_left--;
            if (lit_left == 0 && mat_left) {
                uint32_t window = (uint32_t)(pos + 1) < 32768u ? (uint32_t)(pos + 1) : 32768u;
                dist = 1 + ((lcg(&s) >> 4) % window);
            }
        } else {
            b = out[pos - (int)dist];
            mat_left--;
        }
        out[pos] = b;
    }
}

void lz4s_fill_block(int dist, uint64_t seed, uint64_t block_index, uint8_t* out, int len)
{
    uint64_t key = block_key(seed, block_index);
    switch (dist) {
    case LZ4S_ZEROS:   memset(out, 0, (size_t)len); break;
    case LZ4S_RANDOM:  fill_random(key, out, len); break;
    case LZ4S_FUZ:     fill_fuz(key, out, len); break;
    default:           fill_records(key, out, len); break;
    }
}

void lz4s_fill_batch(int dist, uint64_t seed, uint64_t first_block, int64_t n, uint8_t* out,
                     int64_t stride, int len)
{
    for (int64_t i = 0; i < n; i++) lz4s_fill_block(dist, seed, first_block + (uint64_t)i, out + i * stride, len);
}

uint64_t lz4s_checksum(const uint8_t* p, int64_t n)
{
    uint64_t h = 0;
    for (int64_t w = 0; w * 8 < n; w++) {
        uint64_t v = 0;
        for (int k = 0; k < 8 && w * 8 + k < n; k++) v |= (uint64_t)p[w * 8 + k] << (8 * k);
        h += mix64(v + (uint64_t)w * 0xD1342543DE82EF95ull);
    }
    return h + mix64((uint64_t)(uint32_t)n);
}
