/*
 * lz4hip.h -- C ABI of liblz4hip.so: the MI355X (gfx950) batched LZ4 block codec that stands in for
 * lz4net's back-end ladder (LZ4mm / LZ4cc / LZ4pn / LZ4ps) behind LZ4.LZ4Codec.Encode/Decode/EncodeHC.
 *
 * Plain C, plain pointers and sizes: P/Invoke-able from C# (bindings/csharp/HipLZ4Service.cs), callable
 * through ctypes (lz4net_amd/_lib.py) and from C/C++ (include/lz4net/LZ4Codec.hpp).  All paths below
 * citing the reference are relative to the lz4net repository.
 *
 * Conventions (identical to the reference's core functions):
 *   encode           : bytes written, or 0 when the output limit would be exceeded
 *   decode, known    : bytes CONSUMED from the source, or -(error position in the source)
 *   decode, unknown  : bytes PRODUCED, or -(error position in the source)
 * Library-level failures (no device, HIP error, bad argument) are reported as LZ4HIP_E_* values,
 * which are far outside the range of any codec result, plus lz4hip_last_error().
 * There is no CPU fallback: without a usable gfx950 device every call fails with LZ4HIP_E_DEVICE.
 *
 * Thread safety: all entry points are re-entrant.  Host-pointer calls use per-thread device scratch
 * and the per-thread default stream; device-pointer calls are asynchronous on the stream given.
 */
#ifndef LZ4HIP_H
#define LZ4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4HIP_E_DEVICE   (-2000000001)   /* no usable HIP device / HIP runtime error */
#define LZ4HIP_E_ARGUMENT (-2000000002)   /* null pointer, negative size, ... */
#define LZ4HIP_E_MEMORY   (-2000000003)   /* device allocation failed */

#define LZ4HIP_MODE_FAST 0
#define LZ4HIP_MODE_HC   1

/* ---- information ------------------------------------------------------------------------------ */
/* lz4hip_uncompress is not told its source length (neither is LZ4_uncompress, original/lz4.c:812-814): like the
 * reference it trusts the stream and reads on until `osize` bytes are produced -- on a corrupt stream that walk is
 * unbounded on the host side exactly as the reference's is.  Use lz4hip_uncompress_bounded for untrusted input. */
/* Counterpart of LZ4Codec.CodecName (src/LZ4/LZ4Codec.cs:298-308), e.g. "HIP gfx950 (AMD Instinct MI355X)". */
const char* lz4hip_codec_name(void);
int         lz4hip_device_count(void);
const char* lz4hip_last_error(void);
/* The kernel sources this binary was compiled from: the first 16 hex digits of the SHA-256 over lz4net_amd/csrc/ (file names + contents, sorted;
 * lz4net_amd/build.py csrc_sha()), "+tuning" appended for -DLZ4HIP_TUNING_BUILD libraries, "unknown" for a build that bypassed build.py.
 * bench.py prints it next to the hash of the tree it runs from and refuses to measure when the two differ. */
const char* lz4hip_build_id(void);

/* LZ4_compressBound (original/lz4.h:85-86) == LZ4Codec.MaximumOutputLength (src/LZ4ps/LZ4Codec.cs:142-145). */
int lz4hip_compressBound(int isize);

/* ---- single block, host memory, lz4.h-shaped ----------------------------------------------------
 * Drop-in for the functions lz4net's native back-end binds (src/LZ4cc/LZ4Codec.64.cpp:31-36,81-100,
 * 139-144 call I64_LZ4_compress_limitedOutput / I64_LZ4_uncompress / I64_LZ4_uncompress_unknownOutputSize /
 * I64_LZ4_compressHC_limitedOutput; declarations original/lz4.h:59-60,101,116 and original/lz4hc.h:47,57). */
int lz4hip_compress_limitedOutput(const char* source, char* dest, int isize, int maxOutputSize);
int lz4hip_compress(const char* source, char* dest, int isize);
int lz4hip_compressHC_limitedOutput(const char* source, char* dest, int isize, int maxOutputSize);
int lz4hip_compressHC(const char* source, char* dest, int isize);
int lz4hip_uncompress(const char* source, char* dest, int osize);
int lz4hip_uncompress_unknownOutputSize(const char* source, char* dest, int isize, int maxOutputSize);
/* Known-size decode that is also told the source length and never reads past it -- what
 * ILZ4Service.Decode(..., knownOutputLength: true) needs (src/LZ4/ILZ4Service.cs:30-36: the service
 * always has inputLength; the wrapper compares the result with it, src/LZ4pn/LZ4Codec.Unsafe.cs:373-378). */
int lz4hip_uncompress_bounded(const char* source, int isize, char* dest, int osize);

/* ---- batches -------------------------------------------------------------------------------------
 * The reference has no batch API; lz4net users loop over LZ4Codec.Encode/Decode per block (e.g.
 * LZ4Stream.FlushCurrentChunk / AcquireNextChunk, src/LZ4/LZ4Stream.cs:239-312).  One descriptor
 * describes n independent blocks; block i starts at base + (off ? off[i] : i * stride). */
typedef struct lz4hip_batch {
    const void*    src;
    const int64_t* src_off;      /* optional byte offsets (packed layouts), else NULL */
    int64_t        src_stride;
    const int32_t* src_len;      /* per-block input bytes, or NULL to use src_len_all */
    void*          dst;
    const int64_t* dst_off;
    int64_t        dst_stride;
    const int32_t* dst_cap;      /* per-block capacity (encode, unknown-size decode) or exact size (known-size decode); NULL => dst_cap_all */
    int32_t        dst_cap_all;
    int32_t        src_len_all;  /* length of every block when src_len == NULL; otherwise a HINT: 0 = unknown, else it MUST be an upper
                                    bound on every src_len[i] (LZ4HC picks its 16-bit-head kernels from "<= 65536"; a too-small
                                    value with longer blocks is a caller error and yields LZ4HIP_E_ARGUMENT results for those blocks) */
    int32_t*       result;       /* per-block codec result, conventions above */
    int64_t        n_blocks;
} lz4hip_batch_t;

/* Device-resident batches: every pointer in *b is device memory of the CURRENT device; the call only
 * enqueues kernels on `stream` (a hipStream_t, NULL = default stream) and returns 0 or LZ4HIP_E_*.
 * One exception, once per device and size: the FIRST fast-encode batch of >= 49152 blocks on a device (and a later one that needs
 * more resident wavefronts than any before) builds the lane encoder's table slab inside the call -- device allocations, for
 * slabs >= 2 GiB a few timed probe launches (hipEventSynchronize) on a stream of the library's own, and, when a smaller slab is
 * replaced, one hipDeviceSynchronize before that one is freed (it stays in place if the larger one cannot be had): 0.1 - 3.5 s
 * during which the calling thread blocks and holds the device's encoder workspace.  LZ4HC batches likewise
 * allocate their tables on first use; GROWING them for a later, larger batch waits for the device (hipDeviceSynchronize) before the old tables
 * are freed.  The first lane-mapped decode on a device runs a ~1 ms probe launch and waits for it (knob decoder_wrapped_stores).  Every later call is launch-only. */
int lz4hip_encode_batch_device(const lz4hip_batch_t* b, int mode, void* stream);
int lz4hip_decode_batch_device(const lz4hip_batch_t* b, int known_output_size, void* stream);

/* Host-resident batches: stages through device memory (H2D, kernels, D2H) in slices whose copies and kernels overlap, and synchronises.  DECODE
 * batches of >= 8192 blocks are cut round-robin over two staging pipelines on the current device, each on a persistent worker thread of the
 * library (knob host_workers); fast-ENCODE batches run as one pipeline whose slices are equal and at most one residency round of the
 * wavefront-mapped encoder each (ten blocks per CU): its kernels are the bottleneck. */
int lz4hip_encode_batch_host(const lz4hip_batch_t* b, int mode);
int lz4hip_decode_batch_host(const lz4hip_batch_t* b, int known_output_size);

/* Host-resident batches sharded over several GPUs of the node: block i is processed by the (i mod N)-th device
 * selected by device_mask (bit d = HIP device d; 0 = every visible device) -- the round-robin partition of
 * SURVEY.md 8e.  One PERSISTENT worker thread and one staging pipeline per device (started on first use, reused by every
 * later call; lz4hip_release_workspaces gives their memory back), no inter-device traffic; per-block results
 * and payloads land in the caller's arrays in global block order.  This is what a C# caller (HipLZ4Batch,
 * bindings/csharp) uses to spread LZ4Codec work over the 8 GPUs of a node without any launcher. */
int lz4hip_encode_batch_host_multi(const lz4hip_batch_t* b, int mode, uint64_t device_mask);
int lz4hip_decode_batch_host_multi(const lz4hip_batch_t* b, int known_output_size, uint64_t device_mask);

/* ---- diagnostics ---------------------------------------------------------------------------------
 * Launch counters per kernel family since the library was loaded: which block->hardware mapping a call
 * actually used (the GPU tests assert these).  Copies min(n, LZ4HIP_K_COUNT) counters, returns LZ4HIP_K_COUNT. */
#define LZ4HIP_K_DECODE_WAVE 0   /* lz4hip_decode.hpp:         one wavefront per block */
#define LZ4HIP_K_DECODE_LANE 1   /* lz4hip_decode_lane4.hpp:   one lane per block      */
#define LZ4HIP_K_ENCODE_WAVE 2
#define LZ4HIP_K_ENCODE_LANE 3
#define LZ4HIP_K_HC_WAVE     4
#define LZ4HIP_K_HC_LANE     5
#define LZ4HIP_K_COUNT       6
int lz4hip_dispatch_counts(uint64_t* counts, int n);

/* Named integer knobs for tests, A/B runs and deployment tuning.  Every knob takes its initial value from the
 * environment variable in brackets ONCE, when the library first looks at a knob; the launch paths never call getenv().
 *   "decoder", "encoder", "hc"   [LZ4HIP_DECODER / _ENCODER / _HC = wave | lane]  0 automatic, 1 one wavefront per block,
 *                                 2 one lane per block -- forces that mapping for EVERY block, whatever the batch size
 *   "encoder_waves_per_cu", "hc_waves_per_cu"  [LZ4HIP_ENCODER_WAVES_PER_CU, LZ4HIP_HC_WAVES_PER_CU]  residency of the
 *                                 persistent lane-per-block encoder grids (0 = built-in default)
 *   "hc_groups"                  [LZ4HIP_HC_GROUPS]  wavefronts of the LZ4HC lane grid (0 = from the residency)
 *   "host_threads", "host_slices" [LZ4HIP_HOST_THREADS, LZ4HIP_HOST_SLICES]  host-pointer batches: threads of the process-wide row pool that gathers /
 *                                 scatters rows (0 = a quarter of the hardware threads, at least 8 and at most 64, never more than the host has; read when the pool starts a thread), slices per batch
 *   "decoder_gen", "decoder_ring" [LZ4HIP_DECODER_GEN, LZ4HIP_DECODER_RING]  lane decoder generation (0 default = 4, lz4hip_decode_lane4.hpp;
 *                                 2 and 3 exist only in libraries built with -DLZ4HIP_TUNING_BUILD) and, for generation 4, its
 *                                 configuration: bytes of output ring per lane + 1000 x variant (bit 0: 128-byte flush units,
 *                                 bit 1: 32-byte input pieces, bit 2: one flush store instruction instead of two, bit 3: the flush runs in every second
 *                                 iteration only, bit 4: input pieces are requested in the other iterations only, bit 5: the pieces come out of whole
 *                                 64-byte sectors fetched once; 0 = default 59192; other configurations exist only in tuning builds)
 *   "hc_gen"                     [LZ4HIP_HC_GEN]  LZ4HC lane mapping: 0 default (4 for blocks <= 64 KiB, else 2); 4 the state machine over
 *                                 precomputed chains that carry shared lengths (lz4hip_hc_lcp.hpp), 2 the state machine with the
 *                                 insert loop (lz4hip_hc_conv.hpp: blocks > 64 KiB; smaller ones only in tuning builds); 1 and 3 (one
 *                                 loop nest per lane; precomputed chains without lengths) exist only in -DLZ4HIP_TUNING_BUILD libraries
 *   "hc_ctrl_every", "hc_ctrl_lanes" [LZ4HIP_HC_CTRL_EVERY, LZ4HIP_HC_CTRL_LANES]  generation 4: the parse's control flow runs for all
 *                                 waiting lanes every N-th iteration (a power of two) or as soon as M lanes wait (0 = defaults 8 / 32)
 *   "decoder_persist"            [LZ4HIP_DECODER_PERSIST]  lane decoder: 0 = automatic (one block per lane under hardware dispatch for large batches whose
 *                                 blocks nearly all take the lane mapping; a persistent grid whose lanes pull blocks from a counter for batches of fewer
 *                                 than three residency rounds and for batches with many blocks routed to the wavefront mapping -- counted on the device),
 *                                 1 = always the persistent grid, 2 = never;  "decoder_groups" [LZ4HIP_DECODER_GROUPS]: wavefronts of that grid
 *                                 (0 = the residency; tests use 1 so that every lane restarts hundreds of times)
 *   "hc_sub_chunks"              [LZ4HIP_HC_SUB_CHUNKS]  LZ4HC lane mapping, blocks <= 64 KiB: a chunk of blocks is cut into this many sub-chunks whose
 *                                 table builders and lane kernels overlap on separate streams (0 = default 2, 1 = one after the other, at most 8)
 *   "encoder_slab_tries"         [LZ4HIP_ENCODER_SLAB_TRIES]  fast encoder, lane mapping: its hash tables live in a slab of separately allocated chunks;
 *                                 when the slab is built (first large batch on a device) the library times a probe kernel on it and, if the placement
 *                                 is slower than a well spread one, builds up to this many candidates and keeps the fastest (0 = default 4;
 *                                 1 = the first candidate, unmeasured).  Read-only through lz4hip_tuning_get: "encoder_slab_rate" (what the
 *                                 current device's slab measured, 1000 x G probe steps per second; 0 = none / unmeasured), "encoder_slab_tried" (candidates built),
 *                                 "encoder_slab_chunks" (separate allocations the slab in use consists of)
 *   "host_workers"               [LZ4HIP_HOST_WORKERS]  host-pointer batches of >= 8192 blocks on ONE device (lz4hip_*_batch_host; not LZ4HC) run as this many staging
 *                                 pipelines that share the device -- the persistent workers of the *_multi entry points, block i -> worker i mod k -- so that one pipeline's
 *                                 kernels and copies fill the other's gaps (0 = default: 2 for decode, 1 for fast encode, whose kernels are the bottleneck either way;
 *                                 1 = the calling thread's own pipeline alone; at most 8)
 *   "encoder_wave_version"       [LZ4HIP_ENCODER_WAVE_VERSION]  wavefront-mapped fast encoder, blocks below LZ4_64KLIMIT: 0 = default 2 (encode_fast_block64k, round 6);
 *                                 1 = the first version (exists in -DLZ4HIP_TUNING_BUILD libraries only: A/B runs)
 *   "encoder_wg5"                [LZ4HIP_ENCODER_WG5]  wavefront-mapped fast encoder: 0 = default, workgroups of FIVE blocks (five wavefronts, 80 KiB of LDS; two per CU = ten
 *                                 blocks) wherever that saves the batch a residency round (2 305 - 2 560 blocks, 16 384, everything from 23 040 blocks up on 256 CUs) -- gfx950 hands out LDS in granules of 1 280 bytes, a 16 KiB table takes thirteen of a
 *                                 CU's 128, so only nine one-block workgroups fit; 1 = always one block per workgroup (rounds 1-5); 2 = five per workgroup whatever the size
 *   "decoder_wg4"                [LZ4HIP_DECODER_WG4]  lane decoder, batches of at most one residency round: 0 = default, workgroups of FOUR wavefronts (they go to the four SIMDs of
 *                                 one CU, whatever ran on the device before) while the batch has more than one wavefront per CU and at most one residency round; 1 = always workgroups of one
 *                                 wavefront; 2 = the four-wavefront form from four wavefronts on (tests); 3 = whatever the batch size (A/B runs)
 *   "decoder_wrapped_stores"     [LZ4HIP_DECODER_WRAPPED_STORES]  lane decoder: its default instantiation stores every ring row at `row` and at `row - ring size`
 *                                 and relies on gfx950 dropping the LDS store that falls outside the workgroup's allocation; the library CHECKS that rule once per
 *                                 device before the first lane-mapped decode (a ~1 ms probe launch) and uses the instantiation that wraps its rows instead (same bytes,
 *                                 a few per cent slower) where the probe does not confirm it.  1 = always the wrapped-row instantiation (debuggers, trap-on-violation
 *                                 modes).  Read-only through lz4hip_tuning_get: "decoder_dual_store" = 1 if lane-mapped decodes on the current device store twice, else 0
 *   "logical_devices"            [LZ4HIP_LOGICAL_DEVICES]  the *_multi entry points run this many device workers over the
 *                                 selected devices, wrapping around (0 = one per device): exercises the threaded path on one GPU
 * lz4hip_tuning_set returns the previous value (>= 0) or LZ4HIP_E_ARGUMENT; lz4hip_tuning_get the current value. */
int lz4hip_tuning_set(const char* name, int value);
int lz4hip_tuning_get(const char* name);

/* Frees the grow-only kernel workspaces (encoder hash-table / LZ4HC slabs) of the CURRENT device after
 * waiting for their last user, the calling thread's host-pointer staging (device images, pinned slots) for it, and that of the *_multi entry
 * points' persistent device workers.  The reference frees its tables before every return (original/lz4.c:780-786);
 * the library caches them between calls, this is how a caller gets the memory back. */
int lz4hip_release_workspaces(void);

/* ---- device-side synthetic data + verification (bench / tests) ---------------------
 * dist: 0 zeros, 1 incompressible, 2 reference fuzzer generator (original/fuzzer.c:149-168), 3 record-like.
 * Row i of `out` is synthetic block number first_block + i * block_step (block_step = world size gives a
 * rank its round-robin share of a global batch).  Bit-identical to the CPU twins in oracle/synth.c. */
int lz4hip_synth_device(int dist, uint64_t seed, uint64_t first_block, uint64_t block_step, int64_t n_blocks,
                        void* out, int64_t stride, int32_t len, void* stream);
int lz4hip_checksum_device(const void* data, const int64_t* off, int64_t stride, const int32_t* len,
                           int32_t len_all, uint64_t* sums, int64_t n_blocks, void* stream);
/* *mismatches (device, unsigned 64-bit, caller zeroes it) += number of differing bytes */
int lz4hip_compare_device(const void* a, int64_t a_stride, const void* b, int64_t b_stride,
                          const int32_t* len, int32_t len_all, int64_t n_blocks, uint64_t* mismatches,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LZ4HIP_H */
