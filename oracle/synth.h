/*
 * synth.h -- TEST INFRASTRUCTURE ONLY: CPU twins of the device-side synthetic block generators
 * (lz4net_amd/csrc/lz4hip_synth.hpp).  Both sides must produce bit-identical bytes for the same
 * (distribution, seed, block index, length) so that full-size GPU batches can be spot-checked
 * against the oracle without ever holding the batch in host memory.
 *
 *   D0 zeros           all 0x00
 *   D1 incompressible  counter-based splitmix64 words
 *   D2 fuzzer-style    the reference's own fuzzer generator, original/fuzzer.c:81-85,149-168
 *                      (4 LCG "sequence" seeds, re-seeded at random), seeded per block
 *   D3 record-like     literal runs of 4..27 fresh bytes followed by 8..95-byte copies from up
 *                      to 32 KiB back (log/record-like data: long matches, ratio ~0.3)
 */
#ifndef LZ4_SYNTH_H
#define LZ4_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { LZ4S_ZEROS = 0, LZ4S_RANDOM = 1, LZ4S_FUZ = 2, LZ4S_RECORDS = 3 };
void lz4s_fill_block(int dist, uint64_t seed, uint64_t block_index, uint8_t* out, int len);
/* n blocks, block i written at out + i*stride */
void lz4s_fill_batch(int dist, uint64_t seed, uint64_t first_block, int64_t n, uint8_t* out,
                     int64_t stride, int len);
/* position-salted 64-bit word-sum checksum of a byte range; device twin: csrc/lz4hip_synth.hpp */
uint64_t lz4s_checksum(const uint8_t* p, int64_t n);
#ifdef __cplusplus
}
#endif
#endif
