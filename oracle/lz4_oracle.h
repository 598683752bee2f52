/*
 * lz4_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, index based, exact-length copies) of the LZ4 r93-era block codec that
 * lz4net's LZ4pn/LZ4ps/LZ4cc back-ends all implement (reference: original/lz4.c, original/lz4hc.c;
 * generated C# twins src/LZ4pn/LZ4Codec.Unsafe64*.Dirty.cs).  It exists so that tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() can CHECK the HIP path.  Nothing under lz4net_amd/
 * (the product) may include, link or call it.
 *
 * Parity is PINNED: tests/test_oracle_vs_ref.py compares every function here byte-for-byte and
 * return-code-for-return-code against the reference's own C compiled in place
 * (oracle/_ref/libref_lz4.so, built by oracle/Makefile from /root/reference/original), and
 * tests/golden/ holds vectors generated from that library.
 */
#ifndef LZ4_ORACLE_H
#define LZ4_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* original/lz4.h:85-86  ==  src/LZ4ps/LZ4Codec.cs:142-145 (MaximumOutputLength) */
int lz4o_compress_bound(int isize);

/* original/lz4.c:774-792 dispatcher -> :573-771 (64k variant) or :345-562 (generic variant).
 * Returns bytes written, 0 when the output limit would be exceeded. */
int lz4o_compress_limited(const uint8_t* src, uint8_t* dst, int isize, int max_out);

/* original/lz4.c:812-914.  Returns bytes CONSUMED from src, or -(error position in src). */
int lz4o_uncompress(const uint8_t* src, uint8_t* dst, int osize);

/* original/lz4.c:916-1044. Returns bytes PRODUCED, or -(error position in src). */
int lz4o_uncompress_unknown(const uint8_t* src, uint8_t* dst, int isize, int max_out);

/* original/lz4hc.c:745-755 -> :557-742 with the match finder of :330-518.
 * Returns bytes written, 0 when the output limit would be exceeded. */
int lz4o_compress_hc_limited(const uint8_t* src, uint8_t* dst, int isize, int max_out);

#ifdef __cplusplus
}
#endif
#endif
