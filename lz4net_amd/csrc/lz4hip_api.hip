// lz4hip_api.hip -- host side of liblz4hip.so: the C ABI declared in include/lz4hip.h.
//
// Only launches the gfx950 kernels of this directory; there is deliberately NO CPU code path for
// the codec here (a missing/unsupported device is an error, never a fallback).
#include "lz4hip_wave.hpp"

#include "lz4hip_common.hpp"
#include "lz4hip_decode.hpp"
#include "lz4hip_decode_lane4.hpp"
#ifdef LZ4HIP_TUNING_BUILD            /* superseded generations, for same-box A/B runs only (tools/ab/) */
#include "../../tools/ab/lz4hip_decode_lane.hpp"
#include "../../tools/ab/lz4hip_decode_lane3.hpp"
#include "../../tools/ab/lz4hip_hc_lane.hpp"
#include "../../tools/ab/lz4hip_hc_nat_lane.hpp"
#endif
#include "lz4hip_encode.hpp"
#include "lz4hip_encode_lane.hpp"
#include "lz4hip_hc.hpp"
#include "lz4hip_hc_conv.hpp"
#include "lz4hip_hc_nat.hpp"
#include "lz4hip_hc_lcp.hpp"
#include "lz4hip_synth.hpp"

#include "../../include/lz4hip.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace lz4hip;

namespace {

thread_local std::string g_last_error;

// Launch counters per kernel family (lz4hip_dispatch_counts): what the tests assert so that a test that names a
// mapping is known to have run it.
std::atomic<uint64_t> g_dispatch[LZ4HIP_K_COUNT];
void count_dispatch(int k) { g_dispatch[k].fetch_add(1, std::memory_order_relaxed); }

// Named integer knobs (lz4hip_tuning_set / _get).  Each one takes its initial value from the environment ONCE, the
// first time any knob is looked at; after that the launch paths read an atomic and never call getenv().
// Mappings: 0 = automatic, 1 = one wavefront per block, 2 = one lane per block.
enum Knob { kKnobDecoder = 0, kKnobEncoder, kKnobHc, kKnobEncoderWavesPerCu, kKnobHcWavesPerCu, kKnobHcGroups,
            kKnobHostThreads, kKnobHostSlices, kKnobLogicalDevices, kKnobDecoderGen, kKnobDecoderRing, kKnobHcGen, kKnobHcCtrlEvery, kKnobHcCtrlLanes, kKnobHcSubChunks, kKnobDecoderPersist, kKnobDecoderGroups, kKnobEncoderSlabTries, kKnobEncoderWaveVersion, kKnobEncoderWg5, kKnobHostWorkers, kKnobDecoderWg4, kKnobDecoderWrappedStores, kKnobCount };
struct KnobInfo { const char* name; const char* env; bool mapping; };
const KnobInfo kKnobInfo[kKnobCount] = {
    { "decoder", "LZ4HIP_DECODER", true }, { "encoder", "LZ4HIP_ENCODER", true }, { "hc", "LZ4HIP_HC", true },
    { "encoder_waves_per_cu", "LZ4HIP_ENCODER_WAVES_PER_CU", false }, { "hc_waves_per_cu", "LZ4HIP_HC_WAVES_PER_CU", false },
    { "hc_groups", "LZ4HIP_HC_GROUPS", false },                     // persistent grid of the LZ4HC lane kernel (tests: few lanes, many blocks each)
    { "host_threads", "LZ4HIP_HOST_THREADS", false }, { "host_slices", "LZ4HIP_HOST_SLICES", false },
    { "logical_devices", "LZ4HIP_LOGICAL_DEVICES", false },         // tests: N workers of the multi-device path over the visible devices (wrapping around)
    { "decoder_gen", "LZ4HIP_DECODER_GEN", false },                 // lane decoder: 0 default = 4 lz4hip_decode_lane4.hpp; 2 and 3 (tools/ab/) only in LZ4HIP_TUNING_BUILD libraries
    { "decoder_ring", "LZ4HIP_DECODER_RING", false },               // generation 4: ring bytes + 1000 x variant (0 default; other configurations only in LZ4HIP_TUNING_BUILD libraries)
    { "hc_gen", "LZ4HIP_HC_GEN", false },                           // LZ4HC lane mapping: 0 default; 4 lz4hip_hc_lcp.hpp (blocks <= 64 KiB), 2 lz4hip_hc_conv.hpp (larger blocks; <= 64 KiB in tuning builds); 1 lz4hip_hc_lane.hpp and 3 lz4hip_hc_nat.hpp in tuning builds only
    { "hc_ctrl_every", "LZ4HIP_HC_CTRL_EVERY", false }, { "hc_ctrl_lanes", "LZ4HIP_HC_CTRL_LANES", false },   // lz4hip_hc_lcp.hpp: control-flow batching (0 default)
    { "hc_sub_chunks", "LZ4HIP_HC_SUB_CHUNKS", false },             // LZ4HC lane launch: sub-chunks whose table builders and lane kernels overlap (0 default = 2, 1 = one after the other, max 8)
    { "decoder_persist", "LZ4HIP_DECODER_PERSIST", false },         // lane decoder, default configuration: 0 the device picks one block per lane or the persistent grid, 1 always persistent, 2 never
    { "decoder_groups", "LZ4HIP_DECODER_GROUPS", false },           // tests: wavefronts of the persistent lane decoder's grid (0 = the residency): few lanes, many restarts each
    { "encoder_slab_tries", "LZ4HIP_ENCODER_SLAB_TRIES", false },   // lane encoder's table slab: candidate placements that are built and measured (0 default = 4; 1 = the first one, unmeasured)
    { "encoder_wave_version", "LZ4HIP_ENCODER_WAVE_VERSION", false },   // wavefront-mapped fast encoder, blocks < 64 KiB + 11: 0 default = 2 (encode_fast_block64k); 1 = the first version, in LZ4HIP_TUNING_BUILD libraries only
    { "encoder_wg5", "LZ4HIP_ENCODER_WG5", false },                   // wavefront-mapped fast encoder: 0 default = workgroups of FIVE blocks (80 KiB of LDS: two per CU = ten blocks) wherever that saves a residency round against one-block workgroups (nine per CU: 16 KiB is thirteen of the CU's 128 LDS granules of 1 280 bytes); 1 = always one block per workgroup (rounds 1-5); 2 = five per workgroup whatever the batch size (tests, A/B runs)
    { "host_workers", "LZ4HIP_HOST_WORKERS", false },                 // single-device host-pointer batches of >= 8192 blocks: staging pipelines (persistent worker threads) that share the device, each taking every k-th block (0 default = 2 for decode, 1 for the encoders; 1 = the calling thread's pipeline alone, rounds 2-5)
    { "decoder_wg4", "LZ4HIP_DECODER_WG4", false },                   // lane decoder, batches of at most one residency round: 0 default = workgroups of FOUR wavefronts (one per SIMD of a CU) while the batch has more than one wavefront per CU and at most one residency round; 1 = always workgroups of one wavefront (rounds 1-5); 2 = the four-wavefront form from four wavefronts on (tests); 3 = whatever the batch size (A/B runs)
    { "decoder_wrapped_stores", "LZ4HIP_DECODER_WRAPPED_STORES", false },   // lane decoder: 1 = the instantiation that WRAPS its ring rows (no LDS store outside the allocation) whatever the device's probe said; 0 default = what the probe allows (read-only twin: "decoder_dual_store")
};
std::atomic<int> g_knob[kKnobCount];
std::once_flag g_knob_once;
void knobs_init()
{
    std::call_once(g_knob_once, [] {
        for (int k = 0; k < kKnobCount; k++) {
            const char* e = getenv(kKnobInfo[k].env);
            int v = 0;
            if (e && kKnobInfo[k].mapping) v = e[0] == 'w' ? 1 : (e[0] == 'l' ? 2 : 0);
            else if (e) v = atoi(e);
            g_knob[k].store(v < 0 ? 0 : v, std::memory_order_relaxed);
        }
    });
}
int knob(int k) { knobs_init(); return g_knob[k].load(std::memory_order_relaxed); }

// A lone wavefront of the lane mapping needs milliseconds for its 64 blocks, so the mapping only pays once the
// batch fills the GPU (measured crossover 13 k (D2) .. 28 k (D3) blocks, profiles/r01/decode_small_batches.txt).
constexpr int64_t kLaneDecodeMinBlocks = 16384;
constexpr int64_t kLaneEncodeMinBlocks = 49152;  // a lane needs 60 - 110 ms for its block whatever the batch: the wavefront mapping alone is faster up to ~46 k (D2) /
                                                // ~58 k (D3) blocks since its second version (profiles/r06/encoder_mapping_crossover.txt; round 5: 32768, until then 16384)
constexpr int kLaneDecodeGeneration = 4;
constexpr int kLane4Config = 59192;          // lane decoder: 192-byte ring, 32-byte input pieces out of whole 64-byte sectors, 128-byte flush units; iterations
                                              // alternate between flushing (two store instructions) and requesting input (four load instructions, one sector)
constexpr int64_t kHcHostSliceBlocks = 16384;  // host-pointer LZ4HC batches: blocks per slice
constexpr int kHcLaneGeneration = 4;           // blocks <= 64 KiB; larger ones: 2

int fail(int code, const std::string& what)
{
    g_last_error = what;
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? LZ4HIP_E_MEMORY : LZ4HIP_E_DEVICE,          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

// The library refuses to run anywhere but on the architecture its kernels were written for.
int ensure_device()
{
    static thread_local int checked_device = -1;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(LZ4HIP_E_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (dev == checked_device) return 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(LZ4HIP_E_DEVICE, std::string("liblz4hip is built for gfx950 only, device is ") + prop.gcnArchName);
    checked_device = dev;
    return 0;
}

Batch to_device_batch(const lz4hip_batch_t& b)
{
    Batch d;
    d.src = (const uint8_t*)b.src; d.src_off = b.src_off; d.src_stride = b.src_stride; d.src_len = b.src_len;
    d.dst = (uint8_t*)b.dst; d.dst_off = b.dst_off; d.dst_stride = b.dst_stride; d.dst_cap = b.dst_cap;
    d.dst_cap_all = b.dst_cap_all; d.src_len_all = b.src_len_all; d.result = b.result; d.n_blocks = b.n_blocks;
    return d;
}

int check_batch(const lz4hip_batch_t* b)
{
    if (!b) return fail(LZ4HIP_E_ARGUMENT, "batch descriptor is NULL");
    if (b->n_blocks < 0) return fail(LZ4HIP_E_ARGUMENT, "n_blocks < 0");
    if (b->n_blocks > 0 && (!b->src || !b->dst || !b->result))
        return fail(LZ4HIP_E_ARGUMENT, "src, dst and result must be non-NULL");
    if (b->n_blocks > 0x7FFFFFFF) return fail(LZ4HIP_E_ARGUMENT, "n_blocks too large for one launch");
    return 0;
}

// ---- grow-only per-thread device scratch for the host-pointer entry points --------------------
struct Scratch {
    void* p = nullptr; size_t cap = 0; int dev = -1;
    int reserve(size_t n)
    {
        int dev_now = 0;
        HIP_TRY(hipGetDevice(&dev_now));
        if (p && (dev_now != dev || n > cap)) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (!p) {
            size_t want = n < (1u << 20) ? (1u << 20) : n;
            HIP_TRY(hipMalloc(&p, want));
            cap = want; dev = dev_now;
        }
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    ~Scratch() { /* the HIP runtime may already be gone at thread exit: leak on purpose */ }
};

// Per-device kernel workspaces (hash tables of the lane encoder, LZ4HC heads/chains): grow-only, shared by every
// caller of the device.  Kernels that use one must not overlap, whatever streams they were queued on, so a user
// takes a LEASE: the mutex is held while its work is being queued, its stream first waits for the previous user's
// completion event, and the release records the new one.  (GPU-side ordering only; the host never blocks on it.)
struct HcWorkspace {
    void* p = nullptr; size_t cap = 0;
    hipEvent_t last = nullptr; bool busy = false;
    std::mutex mu;
};
HcWorkspace g_hc_ws[64], g_fast_ws[64];

struct Lease {
    HcWorkspace* w = nullptr;
    std::unique_lock<std::mutex> lock;
    void* p = nullptr;
    hipStream_t stream = nullptr;
    bool queued = false;         // kernels that use the workspace have been queued on `stream`
    std::vector<hipStream_t> side;   // ... and on these streams of the library's own (the LZ4HC sub-chunk pipeline)
    // An error return between the first launch and lease_end() must not leave those kernels unaccounted for: the next user
    // (possibly on another stream) would overwrite tables that are still being read.  Work on the side streams is not behind
    // `stream` yet on that path, so it is waited for here (an error path: blocking the host is acceptable).
    ~Lease()
    {
        if (!(w && lock.owns_lock() && queued)) return;
        for (hipStream_t s : side) (void)hipStreamSynchronize(s);
        if (w->last && hipEventRecord(w->last, stream) == hipSuccess) w->busy = true;
    }
};

// Takes the lock on the device's workspace and makes `stream` wait for its previous user.
int lease_begin(HcWorkspace* pool, hipStream_t stream, Lease& l)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(LZ4HIP_E_DEVICE, "device index out of range");
    l.w = &pool[dev];
    l.stream = stream;
    l.lock = std::unique_lock<std::mutex>(l.w->mu);
    if (!l.w->last) HIP_TRY(hipEventCreateWithFlags(&l.w->last, hipEventDisableTiming));
    if (l.w->busy) HIP_TRY(hipStreamWaitEvent(stream, l.w->last, 0));
    return 0;
}
// Makes the leased workspace at least `bytes` large (nonzero return: could not be allocated; the lease stays valid).
int lease_reserve(Lease& l, size_t bytes)
{
    HcWorkspace& w = *l.w;
    if (w.cap < bytes) {
        if (w.p) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(w.p); w.p = nullptr; w.cap = 0; w.busy = false; }
        if (hipMalloc(&w.p, bytes) != hipSuccess) { (void)hipGetLastError(); w.p = nullptr; return fail(LZ4HIP_E_MEMORY, "workspace allocation failed"); }
        w.cap = bytes;
    }
    l.p = w.p;
    return 0;
}
// Everything queued on `stream` so far is this user's work on the workspace.
int lease_end(Lease& l, hipStream_t stream)
{
    l.queued = false;                                                // (recorded here; nothing left for the destructor)
    HIP_TRY(hipEventRecord(l.w->last, stream));
    l.w->busy = true;
    l.lock.unlock();
    return 0;
}

// The lane encoder's table slab (lz4hip_encode_lane.hpp): equally sized chunks, separate allocations, because the kernel's rate is the
// device's rate of random sector read-modify-writes and that depends on how the slab is spread over device memory (20.5 G steps per second
// inside one contiguous 8 GiB allocation, 25-27 G spread out: profiles/r04/random_sectors_*.txt).  Where an allocation lands cannot be
// asked for, but it can be MEASURED: a candidate set of chunks is built and timed with slab_probe_kernel (a few milliseconds); if it is
// slower than a well spread slab, up to three more are built next to it and the fastest stays (fast_slab_reserve).
// One-time work per device and slab size; protected by the g_fast_ws lease.
struct FastSlab {
    std::vector<void*> chunks;
    unsigned tables_per_chunk = 0;   // a multiple of 64
    int64_t groups = 0;              // wavefronts the slab has tables for
    void* ctl = nullptr;             // device: the work counter (256 bytes), then the chunk pointers
    double probe = 0;                // G steps per second the placement measured (0: not measured)
    int tries = 0;
    int64_t failed_groups = 0;       // the smallest slab that could NOT be allocated, and the free device memory at that moment: a request of that
    size_t failed_free = 0;          // size or more is not tried again until more memory is free (or the workspaces were released)
    void release()
    {
        for (void* c : chunks) (void)hipFree(c);
        chunks.clear();
        if (ctl) { (void)hipFree(ctl); ctl = nullptr; }
        groups = 0; tables_per_chunk = 0; probe = 0; tries = 0; failed_groups = 0; failed_free = 0;
    }
};
FastSlab g_fast_slab[64];
constexpr int kFastSlabMaxChunks = 64;
constexpr size_t kFastSlabCtlBytes = 256 + 8 * kFastSlabMaxChunks;

// One candidate: n_chunks separate allocations of chunk_bytes.
bool fast_slab_candidate(int n_chunks, size_t chunk_bytes, std::vector<void*>& out)
{
    for (int k = 0; k < n_chunks; k++) {
        void* c = nullptr;
        if (hipMalloc(&c, chunk_bytes) != hipSuccess) {
            (void)hipGetLastError();
            for (void* q : out) (void)hipFree(q);
            out.clear();
            return false;
        }
        out.push_back(c);
    }
    return true;
}

// Makes the device's slab hold tables for `groups` wavefronts (nonzero return: could not be allocated; the lease stays valid).
int fast_slab_reserve(Lease& l, int dev, int64_t groups)
{
    FastSlab& fs = g_fast_slab[dev];
    if (fs.groups >= groups && fs.ctl) return 0;
    // (a smaller slab that is already there stays until the larger one is built: a failure leaves the old one in place)
    if (fs.failed_groups > 0 && groups >= fs.failed_groups) {            // this size failed before: not again (multi-GiB hipMallocs) unless memory has been freed since
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if (free_b <= fs.failed_free + ((size_t)256 << 20)) return fail(LZ4HIP_E_MEMORY, "workspace allocation failed (not retried: no more device memory free than when it last failed)");
    }
    auto remember_failure = [&] {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if (fs.failed_groups == 0 || groups < fs.failed_groups) fs.failed_groups = groups;
        fs.failed_free = free_b;
    };
    const size_t tables = (size_t)groups * 64;
    const bool large = tables * (size_t)kLaneTableBytes >= ((size_t)2 << 30);   // below 2 GiB: one chunk, nothing to measure
    const int want_tries = !large ? 1 : (knob(kKnobEncoderSlabTries) > 0 ? knob(kKnobEncoderSlabTries) : 4);
    void* ctl = nullptr;
    if (hipMalloc(&ctl, kFastSlabCtlBytes) != hipSuccess) { (void)hipGetLastError(); return fail(LZ4HIP_E_MEMORY, "workspace allocation failed"); }
    hipStream_t ps = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    unsigned* sink = (unsigned*)((uint8_t*)ctl + 128);
    if (want_tries > 1 && (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) {
        (void)hipGetLastError();
        if (ps) (void)hipStreamDestroy(ps);
        if (e0) (void)hipEventDestroy(e0);
        ps = nullptr;                                                 // (no measurement then: the first candidate stays)
    }
    // Candidates: 16 chunks allocated back to back (64 for the third one).  A candidate that measures below the rate of a well spread slab
    // stays allocated while the next one is built -- so that the next one lands elsewhere -- and all but the best are freed at the end
    // (spacer allocations between the chunks spread further but cost seconds: tools/r04/call24.sh, first version).
    const double good = 24.0 * ((double)tables < 262144.0 ? (double)tables / 262144.0 : 1.0);   // G steps per second (profiles/r04/random_sectors_*.txt: 20.5 in one piece, 25-27 spread)
    std::vector<std::vector<void*>> held;
    std::vector<void*> best;
    unsigned best_tpc = 0;
    double best_rate = -1;
    int tried = 0;
    for (int r = 0; r < want_tries; r++) {
        const int n_chunks = !large ? 1 : (r == 2 ? 64 : 16);
        const unsigned tpc = (unsigned)(((groups + n_chunks - 1) / n_chunks) * 64);
        if (r > 0) {                                                   // a further candidate only while it leaves half of the free memory alone
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
            if ((size_t)n_chunks * tpc * (size_t)kLaneTableBytes > free_b / 2) break;
        }
        std::vector<void*> cand;
        if (!fast_slab_candidate(n_chunks, (size_t)tpc * (size_t)kLaneTableBytes, cand)) break;
        tried++;
        double rate = 0;
        if (ps && want_tries > 1) {
            bool ok = hipMemcpyAsync((uint8_t*)ctl + 256, cand.data(), 8 * cand.size(), hipMemcpyHostToDevice, ps) == hipSuccess
                   && hipStreamSynchronize(ps) == hipSuccess;
            const int steps = 300;
            if (ok) {
                hipLaunchKernelGGL(slab_probe_kernel, dim3((unsigned)groups), dim3(64), 0, ps, (uint8_t* const*)((uint8_t*)ctl + 256), tpc, 30, sink);   // (warm-up)
                ok = hipEventRecord(e0, ps) == hipSuccess;
                hipLaunchKernelGGL(slab_probe_kernel, dim3((unsigned)groups), dim3(64), 0, ps, (uint8_t* const*)((uint8_t*)ctl + 256), tpc, steps, sink);
                ok = ok && hipEventRecord(e1, ps) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
                float ms = 0;
                if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) rate = (double)tables * steps / ms / 1e6;
            }
            if (!ok) (void)hipGetLastError();
        }
        if (rate > best_rate) { best.swap(cand); best_tpc = tpc; best_rate = rate; }   // (cand now holds the set that lost)
        if (!cand.empty()) held.push_back(std::move(cand));
        if (rate >= good || rate == 0) break;
    }
    for (auto& h : held) for (void* c : h) (void)hipFree(c);
    if (ps) { (void)hipStreamDestroy(ps); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    if (best.empty()) { (void)hipFree(ctl); remember_failure(); return fail(LZ4HIP_E_MEMORY, "workspace allocation failed"); }
    if (hipMemcpy((uint8_t*)ctl + 256, best.data(), 8 * best.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        for (void* c : best) (void)hipFree(c);
        (void)hipFree(ctl);
        return fail(LZ4HIP_E_DEVICE, "workspace set-up failed");
    }
    if (fs.ctl) {                                                    // the old, smaller slab: its last user first
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        fs.release();
        l.w->busy = false;
    }
    fs.chunks.swap(best); fs.tables_per_chunk = best_tpc; fs.groups = groups; fs.ctl = ctl; fs.probe = best_rate > 0 ? best_rate : 0; fs.tries = tried;
    fs.failed_groups = 0; fs.failed_free = 0;
    return 0;
}

// Streams and events of the pipelined LZ4HC lane launch (one set per device, created on first use; the workspace lease
// serialises its users).
constexpr int kHcSubChunks = 2, kHcMaxSubChunks = 8;   // measured at 2^18 blocks (profiles/r04/hc_sub_chunks.txt): 1 -> 15.9 / 20.8 GB/s (D2 / D3), 2 -> 16.8 / 23.2, 4 -> 15.1 / 21.0, 8 -> 11.6 / 16.7
constexpr size_t kHcCounterBytes = 256 * kHcMaxSubChunks;
struct HcPipe {
    bool ready = false;
    hipStream_t build = nullptr, lane[kHcMaxSubChunks] = {};
    hipEvent_t start = nullptr, built[kHcMaxSubChunks] = {}, done[kHcMaxSubChunks] = {};
    int init()
    {
        if (ready) return 0;
        HIP_TRY(hipStreamCreateWithFlags(&build, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&start, hipEventDisableTiming));
        for (int k = 0; k < kHcMaxSubChunks; k++) {
            HIP_TRY(hipStreamCreateWithFlags(&lane[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&built[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
        }
        ready = true;
        return 0;
    }
    // (lz4hip_release_workspaces, under the LZ4HC workspace's lock, after its last user's event)
    void release()
    {
        if (build) { (void)hipStreamDestroy(build); build = nullptr; }
        if (start) { (void)hipEventDestroy(start); start = nullptr; }
        for (int k = 0; k < kHcMaxSubChunks; k++) {
            if (lane[k]) { (void)hipStreamDestroy(lane[k]); lane[k] = nullptr; }
            if (built[k]) { (void)hipEventDestroy(built[k]); built[k] = nullptr; }
            if (done[k]) { (void)hipEventDestroy(done[k]); done[k] = nullptr; }
        }
        ready = false;
    }
};
HcPipe g_hc_pipe[64];

// Whether the wavefront-mapped fast encoder runs as workgroups of kEncodeBlocksPerGroup blocks (ten blocks per CU) instead of one-block workgroups
// (nine per CU): whenever that saves a residency round -- 2 305 ... 2 560 blocks on 256 CUs, 16 384 (7 rounds instead of 8), every batch from 90 blocks
// per CU up.  With the same number of rounds one block per workgroup is the better form (finer placement, nine wavefronts per CU disturb one another
// less than ten): profiles/r06/wave_encoder_round_steps.txt.
constexpr int kEncodeBlocksPerGroup = 5;
bool encoder_five_blocks_per_workgroup(int64_t n_blocks)
{
    const int k = knob(kKnobEncoderWg5);
    if (k == 1 || n_blocks < kEncodeBlocksPerGroup) return false;
    if (k == 2) return true;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return false;
    const int64_t rounds9 = (n_blocks + 9 * cus - 1) / (9 * cus), rounds10 = (n_blocks + 10 * cus - 1) / (10 * cus);
    return rounds10 < rounds9;
}

int launch_encode(const lz4hip_batch_t* b, int mode, hipStream_t stream)
{
    if (b->n_blocks == 0) return 0;
    const Batch d = to_device_batch(*b);
    if (mode == LZ4HIP_MODE_FAST) {
        // Two mappings (lz4hip_encode.hpp: one wavefront per block, table in LDS, 64 probes of the match search per
        // step; lz4hip_encode_lane.hpp: one lane per block, tables in a global slab).  The first is several times
        // faster where matches are far between (incompressible data: 6x), the second where sequences are short (a
        // chain of dependent steps per sequence: 64 chains per wavefront instead of one).  A batch large enough to
        // fill the lanes is therefore encoded by two launches: the wavefront mapping over every block, which hands a
        // block over (kDeferredResult) as soon as 16 consecutive sequences cover less than 1 KiB, then the lane mapping
        // over the blocks handed over.  Small batches use the wavefront mapping only.
        // The "encoder" knob (LZ4HIP_ENCODER=wave|lane at load time, lz4hip_tuning_set) forces ONE mapping for every block.
        const int force = knob(kKnobEncoder);
        char pick = d.n_blocks >= kLaneEncodeMinBlocks ? 'a' : 'w';  // 'a': both launches
        if (force) pick = force == 1 ? 'w' : 'l';
        Lease lease;
        void* ws = nullptr;                                           // the slab's control block: work counter, then the chunk pointers
        unsigned slab_tpc = 0;
        int64_t groups = 0;
        if (pick != 'w') {
            int dev = 0, cus = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            int wpc = knob(kKnobEncoderWavesPerCu) > 0 ? knob(kKnobEncoderWavesPerCu) : kLaneEncodeWavesPerCu;
            int rc = lease_begin(g_fast_ws, stream, lease);
            if (rc) return rc;
            // the slab holds one table per resident lane; if it cannot be had, halve the residency, and
            // in the end fall back to the wavefront mapping (which needs no workspace)
            if (dev < 0 || dev >= 64) return fail(LZ4HIP_E_DEVICE, "device index out of range");
            for (; wpc >= 1; wpc /= 2) {
                groups = (int64_t)cus * wpc;
                if (groups > (d.n_blocks + 63) / 64) groups = (d.n_blocks + 63) / 64;
                if (fast_slab_reserve(lease, dev, groups) == 0) { ws = g_fast_slab[dev].ctl; slab_tpc = g_fast_slab[dev].tables_per_chunk; break; }
            }
            if (!ws) { lease.lock.unlock(); pick = 'w'; }
        }
        if (pick != 'l') {
            bool first_version = false;
#ifdef LZ4HIP_TUNING_BUILD
            first_version = knob(kKnobEncoderWaveVersion) == 1;
            if (first_version)
                hipLaunchKernelGGL(encode_fast_kernel<1>, dim3((unsigned)d.n_blocks), dim3(64), kFastTableBytes, stream, d, pick == 'a' ? (int)kEncodeMayDefer : 0);
#endif
            // gfx950 hands out LDS in granules of 1 280 bytes, 128 per CU: a 16 KiB table takes thirteen, so NINE one-block workgroups fit a CU;
            // five tables are exactly 64 granules, two such workgroups fill the CU with TEN blocks (profiles/r06/wave_encoder_round_steps.txt).
            // 80 KiB of dynamic LDS has to be allowed per device first; where that is refused the one-block form runs.
            bool five = !first_version && encoder_five_blocks_per_workgroup(d.n_blocks);
            if (five) {
                static std::atomic<int> attr_state[64];                  // 0 not asked yet, 1 allowed, 2 refused
                int dev5 = 0;
                HIP_TRY(hipGetDevice(&dev5));
                if (dev5 < 0 || dev5 >= 64) return fail(LZ4HIP_E_DEVICE, "device index out of range");
                int st = attr_state[dev5].load(std::memory_order_acquire);
                if (st == 0) {
                    st = hipFuncSetAttribute((const void*)(encode_fast_kernel<2, kEncodeBlocksPerGroup>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kEncodeBlocksPerGroup * kFastTableBytes) == hipSuccess ? 1 : 2;
                    if (st == 2) (void)hipGetLastError();
                    attr_state[dev5].store(st, std::memory_order_release);
                }
                five = st == 1;
            }
            if (five)
                hipLaunchKernelGGL((encode_fast_kernel<2, kEncodeBlocksPerGroup>), dim3((unsigned)((d.n_blocks + kEncodeBlocksPerGroup - 1) / kEncodeBlocksPerGroup)),
                                   dim3(64 * kEncodeBlocksPerGroup), kEncodeBlocksPerGroup * kFastTableBytes, stream, d, pick == 'a' ? (int)kEncodeMayDefer : 0);
            else if (!first_version)
                hipLaunchKernelGGL(encode_fast_kernel<2>, dim3((unsigned)d.n_blocks), dim3(64), kFastTableBytes, stream, d, pick == 'a' ? (int)kEncodeMayDefer : 0);
            HIP_TRY(hipGetLastError());
            count_dispatch(LZ4HIP_K_ENCODE_WAVE);
        }
        if (pick != 'w') {
            lease.queued = true;
            HIP_TRY(hipMemsetAsync(ws, 0, 256, stream));
            hipLaunchKernelGGL(encode_fast_lane_kernel, dim3((unsigned)groups), dim3(64), 0, stream, d,
                               (unsigned long long*)ws, (uint8_t* const*)((uint8_t*)ws + 256), slab_tpc, pick == 'a' ? 1 : 0);
            HIP_TRY(hipGetLastError());
            count_dispatch(LZ4HIP_K_ENCODE_LANE);
            int rc = lease_end(lease, stream);
            if (rc) return rc;
        }
    } else if (mode == LZ4HIP_MODE_HC) {
        int dev = 0, cus = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        // 16-bit heads (64 KiB of LDS, two workgroups per CU) when every block is known to be <= 64 KiB:
        // uniform length, or per-block lengths with src_len_all carrying an upper bound (0 = unknown).
        const bool small = b->src_len_all > 0 && b->src_len_all <= 65536;
        // Large batches: one lane per block -- blocks <= 64 KiB over tables built up front for a chunk of the batch
        // (lz4hip_hc_lcp.hpp: chain + shared length per position), larger blocks with the insert loop and a per-lane global slab
        // (lz4hip_hc_conv.hpp); if the workspace cannot be allocated, or the batch is small, one wavefront per block (lz4hip_hc.hpp).
        // The "hc" knob (LZ4HIP_HC=wave|lane at load time, lz4hip_tuning_set) overrides.
        const int force = knob(kKnobHc);
        // (a lane-mapped wavefront needs ~0.35 s for its 64 blocks <= 64 KiB however small the batch, the wavefront mapping does
        //  ~11 k blocks per second: measured crossover at 4096 blocks, profiles/r03/hc_small_batches.txt; blocks > 64 KiB take
        //  the kernel with the insert loop, 1.3 - 2.4 s: 16 k blocks, profiles/r01/hc_small_batches.txt)
        bool lane_per_block = d.n_blocks >= (small ? 4096 : 16384);
        if (force == 1) lane_per_block = false;
        if (force == 2) lane_per_block = true;
        Lease lease;
        int rc = lease_begin(g_hc_ws, stream, lease);
        if (rc) return rc;
        if (lane_per_block) {
            int hc_gen = knob(kKnobHcGen) ? knob(kKnobHcGen) : kHcLaneGeneration;
            if (hc_gen >= 3 && !small) hc_gen = 2;                    // lz4hip_hc_nat.hpp / lz4hip_hc_lcp.hpp are for blocks <= 64 KiB
            const size_t slab = small ? kHcLaneSlab16 : kHcLaneSlab32;
            int wpc = knob(kKnobHcWavesPerCu) > 0 ? knob(kKnobHcWavesPerCu) : kHcLaneWavesPerCu;
            void* ws = nullptr;
            int64_t groups = 0, chunk = 0;
            for (; wpc >= 1; wpc /= 2) {
                groups = (int64_t)cus * wpc;
                if (knob(kKnobHcGroups) > 0) groups = knob(kKnobHcGroups);
                if (groups > (d.n_blocks + 63) / 64) groups = (d.n_blocks + 63) / 64;
                // generation 3: one chain table per block of a chunk; a chunk = at most one block per resident lane (but at least
                // 4096), the batch cut into equal chunks (a short last chunk would run at a fraction of the residency)
                chunk = groups * 64 < 4096 ? 4096 : groups * 64;
                const int64_t n_chunks = (d.n_blocks + chunk - 1) / chunk;
                chunk = ((d.n_blocks + n_chunks - 1) / n_chunks + 63) / 64 * 64;
                if (hc_gen >= 3) {
                    // tables: 256 KiB (generation 4) per block of a chunk, rebuilt chunk after chunk -- never more than half of what the
                    // device has free (a 2^18-block chunk is 64 GiB), and a chunk that cannot be had is halved, down to 4096 blocks
                    const size_t entry = hc_gen == 4 ? kHcLcpTableBytes : kHcNatChainBytes;
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
                    const size_t budget = (free_b + lease.w->cap) / 2;
                    int64_t c = chunk;
                    while (c > 4096 && (size_t)c * entry > budget) c = (c / 2 + 63) / 64 * 64;
                    for (;;) {
                        if (lease_reserve(lease, (size_t)c * entry + kHcCounterBytes) == 0) { ws = lease.p; chunk = c; break; }
                        if (c <= 4096) break;
                        c = (c / 2 + 63) / 64 * 64;
                    }
                    if (ws || knob(kKnobHcGroups) > 0) break;
                    continue;
                }
                const size_t bytes = (size_t)groups * 64 * slab;
                if (lease_reserve(lease, bytes + kHcCounterBytes) == 0) { ws = lease.p; break; }
            }
            if (ws && hc_gen >= 3) {
                // lz4hip_hc_nat.hpp / lz4hip_hc_lcp.hpp: per chunk, the table builders (one workgroup per block), then the lane kernel.
                // The builders need LDS and few registers, the lane kernel registers and no LDS, so they overlap well -- but one
                // after the other on one stream they did not overlap at all (round 3: 190 ms of builders in front of 850 ms of lane
                // kernel).  A chunk is therefore cut into up to kHcSubChunks SUB-CHUNKS: the builders run through them on a build
                // stream, and the lane kernel of a sub-chunk starts on its own stream as soon as its tables are there -- next to the
                // builders of the following ones and to the lane kernels of the earlier ones (each takes its share of the persistent
                // grid).  Same tables, same bytes; the table memory is what one chunk needs, as before.
                if (hc_gen == 4) {
                    static std::atomic<bool> attr_set[64];            // once per device, not per launch
                    if (!attr_set[dev].load(std::memory_order_acquire)) {
                        HIP_TRY(hipFuncSetAttribute((const void*)hc_lcp_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kHcLcpFillLdsBytes));
                        attr_set[dev].store(true, std::memory_order_release);
                    }
                }
                HcPipe& hp = g_hc_pipe[dev];
                if ((rc = hp.init())) return rc;
                const size_t entry = hc_gen == 4 ? kHcLcpTableBytes : kHcNatChainBytes;
                uint8_t* const tables = (uint8_t*)ws + kHcCounterBytes;
                lease.queued = true;
                lease.side.assign(hp.lane, hp.lane + kHcMaxSubChunks);
                lease.side.push_back(hp.build);
                HIP_TRY(hipMemsetAsync(ws, 0, kHcCounterBytes, stream));          // one work counter per sub-chunk (256 bytes apart)
                for (int64_t first = 0; first < d.n_blocks; first += chunk) {
                    const int64_t cnt = d.n_blocks - first < chunk ? d.n_blocks - first : chunk;
                    int subs = knob(kKnobHcSubChunks) > 0 ? knob(kKnobHcSubChunks) : kHcSubChunks;
                    subs = subs > kHcMaxSubChunks ? kHcMaxSubChunks : subs;
                    while (subs > 1 && cnt / subs < 4096) subs--;              // (a lane-mapped launch of fewer blocks is all latency)
                    const int64_t per = ((cnt + subs - 1) / subs + 63) / 64 * 64;
                    // everything queued on the caller's stream so far (the previous chunk's lane kernels included: they read the
                    // tables that are rebuilt now, and the counters that are zeroed again) comes first
                    if (first > 0) HIP_TRY(hipMemsetAsync(ws, 0, kHcCounterBytes, stream));
                    HIP_TRY(hipEventRecord(hp.start, stream));
                    HIP_TRY(hipStreamWaitEvent(hp.build, hp.start, 0));
                    int used = 0;
                    for (int64_t s0 = 0; s0 < cnt; s0 += per, used++) {
                        const int64_t sc = cnt - s0 < per ? cnt - s0 : per;
                        uint8_t* const tab = tables + (size_t)s0 * entry;
                        hipStream_t ls = subs > 1 ? hp.lane[used] : stream;
                        hipStream_t bs = subs > 1 ? hp.build : stream;
                        if (hc_gen == 4) {
                            hipLaunchKernelGGL(hc_nat_chain_kernel<uint32_t>, dim3((unsigned)sc), dim3(kHcNatChainThreads), kHcNatLdsBytes, bs, d, (long long)(first + s0), tab);
                            HIP_TRY(hipGetLastError());
                            hipLaunchKernelGGL(hc_lcp_fill_kernel, dim3((unsigned)sc), dim3(kHcLcpFillThreads), kHcLcpFillLdsBytes, bs, d, (long long)(first + s0), tab);
                            HIP_TRY(hipGetLastError());
                        } else {
#ifdef LZ4HIP_TUNING_BUILD
                            hipLaunchKernelGGL(hc_nat_chain_kernel<uint16_t>, dim3((unsigned)sc), dim3(kHcNatChainThreads), kHcNatLdsBytes, bs, d, (long long)(first + s0), tab);
                            HIP_TRY(hipGetLastError());
#else
                            return fail(LZ4HIP_E_ARGUMENT, "hc_gen: this library has no LZ4HC lane kernel of that generation");
#endif
                        }
                        if (subs > 1) {
                            HIP_TRY(hipEventRecord(hp.built[used], hp.build));
                            HIP_TRY(hipStreamWaitEvent(ls, hp.built[used], 0));
                            HIP_TRY(hipStreamWaitEvent(ls, hp.start, 0));        // (the counters' memset, the caller's earlier work)
                        }
                        int64_t g = (sc + 63) / 64 < groups ? (sc + 63) / 64 : groups;
                        unsigned long long* const counter = (unsigned long long*)((uint8_t*)ws + 256 * (size_t)used);
                        if (hc_gen == 4) {
                            int every = knob(kKnobHcCtrlEvery) > 0 ? knob(kKnobHcCtrlEvery) : kHcLcpCtrlEvery;
                            while (every & (every - 1)) every &= every - 1;          // (a power of two)
                            const int lanes = knob(kKnobHcCtrlLanes) > 0 ? knob(kKnobHcCtrlLanes) : kHcLcpCtrlLanes;
                            hipLaunchKernelGGL(encode_hc_lcp_kernel, dim3((unsigned)g), dim3(64), 0, ls, d, (long long)(first + s0), (long long)sc,
                                               counter, tab, every, lanes);
                        }
#ifdef LZ4HIP_TUNING_BUILD
                        else hipLaunchKernelGGL(encode_hc_nat_kernel, dim3((unsigned)g), dim3(64), 0, ls, d, (long long)(first + s0), (long long)sc, counter, tab);
#endif
                        HIP_TRY(hipGetLastError());
                        if (subs > 1) {
                            HIP_TRY(hipEventRecord(hp.done[used], ls));
                            HIP_TRY(hipStreamWaitEvent(stream, hp.done[used], 0));   // the caller's stream ends up behind every sub-chunk
                        }
                    }
                }
                count_dispatch(LZ4HIP_K_HC_LANE);
                return lease_end(lease, stream);
            }
            if (ws) {
                lease.queued = true;
                HIP_TRY(hipMemsetAsync(ws, 0, 256, stream));
                if (hc_gen == 2 && !small)
                    hipLaunchKernelGGL(encode_hc_conv_kernel<uint32_t>, dim3((unsigned)groups), dim3(64), 0, stream, d,
                                       (unsigned long long*)ws, (uint8_t*)ws + 256, (unsigned long long)slab);
#ifdef LZ4HIP_TUNING_BUILD                                              /* the kernels generation 3 replaced, for A/B runs (tools/hc_gen_ab.py) */
                else if (hc_gen == 2)
                    hipLaunchKernelGGL(encode_hc_conv_kernel<uint16_t>, dim3((unsigned)groups), dim3(64), 0, stream, d,
                                       (unsigned long long*)ws, (uint8_t*)ws + 256, (unsigned long long)slab);
                else if (hc_gen == 1)
                    hipLaunchKernelGGL(encode_hc_lane_kernel, dim3((unsigned)groups), dim3(64), 0, stream, d,
                                       (unsigned long long*)ws, (uint8_t*)ws + 256, (unsigned long long)slab);
#endif
                else return fail(LZ4HIP_E_ARGUMENT, "hc_gen: this library has no LZ4HC lane kernel of that generation");
                HIP_TRY(hipGetLastError());
                count_dispatch(LZ4HIP_K_HC_LANE);
                return lease_end(lease, stream);
            }
        }
        const int lds_bytes = small ? kHcLdsHeads16 : kHcLdsHeads32;
        int64_t groups = (int64_t)cus * (small ? kHcGroupsPerCu : 1);
        if (groups > d.n_blocks) groups = d.n_blocks;
        if ((rc = lease_reserve(lease, (size_t)groups * kHcGlobalBytesPerGroup + 256))) return rc;
        void* ws = lease.p;
        // first 8 bytes of the workspace: the work counter of the persistent grid
        lease.queued = true;
        HIP_TRY(hipMemsetAsync(ws, 0, 256, stream));
        if (!small) {
            static std::atomic<bool> attr_set[64];                    // once per device, not per launch
            if (!attr_set[dev].load(std::memory_order_acquire)) {
                HIP_TRY(hipFuncSetAttribute((const void*)encode_hc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
                attr_set[dev].store(true, std::memory_order_release);
            }
        }
        hipLaunchKernelGGL(encode_hc_kernel, dim3((unsigned)groups), dim3(64), lds_bytes, stream, d,
                           (unsigned long long*)ws, (uint8_t*)ws + 256, lds_bytes);
        HIP_TRY(hipGetLastError());
        count_dispatch(LZ4HIP_K_HC_WAVE);
        if ((rc = lease_end(lease, stream))) return rc;
    } else {
        return fail(LZ4HIP_E_ARGUMENT, "mode must be LZ4HIP_MODE_FAST or LZ4HIP_MODE_HC");
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The lane decoder's dual ring stores (lz4hip_decode_lane4.hpp, L4_APPEND) rest on one hardware rule: a DS store outside the workgroup's LDS
// allocation is dropped.  That rule is CHECKED here, once per device, before the first lane-mapped decode (lds_drop_probe_kernel: every CU
// full of workgroups of the decoder's LDS size issuing the decoder's kinds of out-of-range stores; ~1 ms); a device that does not confirm it --
// or a caller that sets the knob decoder_wrapped_stores -- gets the instantiation whose ring rows are wrapped (POL bit 5), same bytes, ~2 % slower.
std::atomic<int> g_lds_drop_state[64];       // 0 not probed yet, 1 confirmed, 2 not confirmed (or the probe could not run)
std::mutex g_lds_drop_mu;
bool lds_drop_confirmed(int dev)
{
    if (dev < 0 || dev >= 64) return false;
    int st = g_lds_drop_state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        std::lock_guard<std::mutex> lk(g_lds_drop_mu);
        st = g_lds_drop_state[dev].load(std::memory_order_acquire);
        if (st == 0) {
            st = 2;
            int cus = 0;
            unsigned* d = nullptr;
            unsigned h = ~0u;
            hipStream_t ps = nullptr;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
                hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d, 256) == hipSuccess) {
                bool ok = hipMemsetAsync(d, 0, 256, ps) == hipSuccess;
                if (ok) {
                    hipLaunchKernelGGL(lds_drop_probe_kernel, dim3((unsigned)cus * 24u), dim3(64), 0, ps, d, 48);
                    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, ps) == hipSuccess && hipStreamSynchronize(ps) == hipSuccess;
                }
                if (ok && h == 0u) st = 1;
            }
            (void)hipGetLastError();
            if (d) (void)hipFree(d);
            if (ps) (void)hipStreamDestroy(ps);
            g_lds_drop_state[dev].store(st, std::memory_order_release);
        }
    }
    return st == 1;
}

// Work counters of the persistent lane decoder: a small per-device ring of 256-byte slots, one per launch in flight.  A slot comes
// back into use after 64 further launches on that device, possibly on another stream, so each slot carries an event that its
// user records after the last kernel that reads it: the next user's stream waits for that event before it zeroes the slot, and
// the slot's mutex is held from the zeroing to the record (a second host thread that draws the same slot waits for the
// first one's record, not just for its launch).  GPU-side ordering only; a stream never waits unless 64 launches are in flight.
struct CounterSlot { std::mutex mu; hipEvent_t done = nullptr; bool used = false; };
struct CounterRing { void* mem = nullptr; CounterSlot slot[64]; std::atomic<unsigned> next{ 0 }; std::mutex mu; };
CounterRing g_counter_ring[64];
struct CounterLease {
    CounterSlot* s = nullptr;
    std::unique_lock<std::mutex> lock;
    hipStream_t stream = nullptr;
    // (also on error returns: whatever was queued on `stream` so far is what may still touch the slot)
    ~CounterLease() { if (s && lock.owns_lock() && s->done && hipEventRecord(s->done, stream) == hipSuccess) s->used = true; }
};
int decoder_counter(int dev, hipStream_t stream, unsigned long long** out, CounterLease& lease)
{
    if (dev < 0 || dev >= 64) return fail(LZ4HIP_E_DEVICE, "device index out of range");
    CounterRing& r = g_counter_ring[dev];
    void* mem = nullptr;
    unsigned k = 0;
    {
        // the slot's lock is taken while the ring's is still held: lz4hip_release_workspaces() (ring lock, then every slot's lock) can then
        // neither free the ring between the two nor destroy the slot's event before this user has recorded it
        std::lock_guard<std::mutex> lk(r.mu);
        if (!r.mem) HIP_TRY(hipMalloc(&r.mem, 64 * 256));
        mem = r.mem;
        k = r.next.fetch_add(1) & 63u;
        lease.s = &r.slot[k];
        lease.stream = stream;
        lease.lock = std::unique_lock<std::mutex>(lease.s->mu);
        if (!lease.s->done) HIP_TRY(hipEventCreateWithFlags(&lease.s->done, hipEventDisableTiming));
    }
    if (lease.s->used) HIP_TRY(hipStreamWaitEvent(stream, lease.s->done, 0));
    uint8_t* slot = (uint8_t*)mem + 256 * (size_t)k;
    HIP_TRY(hipMemsetAsync(slot, 0, 16, stream));                   // work counter + selected-block count
    *out = (unsigned long long*)slot;
    return 0;
}

int launch_decode(const lz4hip_batch_t* b, int known, hipStream_t stream)
{
    if (b->n_blocks == 0) return 0;
    const Batch d = to_device_batch(*b);
    // Two mappings of the same decoder (lz4hip_decode.hpp: one wavefront per block, coalesced wide copies;
    // lz4hip_decode_lane4.hpp: one lane per block, 64 blocks in flight per wavefront).  A batch is
    // partitioned per block by block_selected(): two launches, each skipping the other's blocks.
    // Small batches cannot fill the lanes and use the wavefront mapping only.
    // The "decoder" knob (LZ4HIP_DECODER=wave|lane at load time, lz4hip_tuning_set) forces one mapping for EVERY block,
    // whatever the batch size (tests, A-B runs).
    const int force = knob(kKnobDecoder);
    int wave_filter = kStreamingBlocks, lane_filter = kFineGrainedBlocks;
    if (force == 1) { wave_filter = kAllBlocks; lane_filter = -1; }
    else if (force == 2) { lane_filter = kAllBlocks; wave_filter = -1; }
    else if (d.n_blocks < kLaneDecodeMinBlocks) { wave_filter = kAllBlocks; lane_filter = -1; }
    if (lane_filter >= 0) {
        const unsigned grid = (unsigned)((d.n_blocks + 63) / 64);
        // Generation 4 (lz4hip_decode_lane4.hpp): LDS per wavefront = 64 x ring + 256..512 bytes decides the residency (ring 192:
        // 12 wavefronts per CU); the configurations that were measured are in profiles/r04/decoder_gen4_*.txt.
        const int gen = knob(kKnobDecoderGen) ? knob(kKnobDecoderGen) : kLaneDecodeGeneration;
        if (gen == 4) {
            // decoder_ring = ring bytes + 1000 x variant (variant bit 0: 128-byte flush units, bit 1: 32-byte input pieces,
            // bit 2: one flush store instruction per iteration, bit 3: the flush runs in every second iteration only,
            // bit 4: input pieces are requested in the other iterations only, bit 5: sector input -- L holds a whole 64-byte sector
            // of the source and feeds the window one 32-byte half at a time)
            const int cfg = knob(kKnobDecoderRing) ? knob(kKnobDecoderRing) : kLane4Config;
            int dev = 0, cus = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            // ring rows stored twice (out-of-range LDS stores dropped by the hardware) only where this device's probe confirmed the rule
            const bool wrapped = knob(kKnobDecoderWrappedStores) != 0 || !lds_drop_confirmed(dev);
#define LZ4HIP_LAUNCH_LANE4_POL(RING, PIECE, FLUSH, FS, FE, IE, POL)                                                                \
            do {                                                                                                                \
                if (known) hipLaunchKernelGGL((decode_lane4_kernel<true, RING, PIECE, FLUSH, FS, FE, IE, POL>), dim3(grid), dim3(64), 0, stream, d, lane_filter);  \
                else       hipLaunchKernelGGL((decode_lane4_kernel<false, RING, PIECE, FLUSH, FS, FE, IE, POL>), dim3(grid), dim3(64), 0, stream, d, lane_filter); \
            } while (0)
#define LZ4HIP_LAUNCH_LANE4(RING, PIECE, FLUSH, FS, FE, IE, POL)                                                                    \
            do {                                                                                                                \
                if (wrapped) LZ4HIP_LAUNCH_LANE4_POL(RING, PIECE, FLUSH, FS, FE, IE, (POL) | 32);                                   \
                else         LZ4HIP_LAUNCH_LANE4_POL(RING, PIECE, FLUSH, FS, FE, IE, POL);                                          \
            } while (0)
#define LZ4HIP_LANE4_CASE(CFG) case CFG: LZ4HIP_LAUNCH_LANE4((CFG) % 1000, ((CFG) / 1000 & 2) ? 32 : 64, ((CFG) / 1000 & 1) ? 128 : 64, ((CFG) / 1000 & 4) ? 1 : 2, ((CFG) / 1000 & 8) ? 2 : 1, ((CFG) / 1000 & 16) ? 2 : 1, ((CFG) / 1000 & 32) ? 16 : 0); break
            // Default configuration: TWO forms of the same kernel.  One block per lane under hardware dispatch is the faster one for
            // a large batch whose blocks all take the lane mapping (2^20 D2 blocks: 980 vs 952 GB/s); the persistent grid, whose lanes pull
            // blocks from a counter and skip what the filter does not select, wins when a wavefront would idle otherwise -- a batch of
            // more than one but fewer than three residency rounds (2^18 blocks: 865 -> 938 GB/s) or a batch with many blocks routed to the
            // wavefront mapping (half zeros: 68 -> 44 ms).  Which one runs is decided ON THE DEVICE: a counting launch, then both kernels, each of which
            // returns at once unless the count says it is its turn (profiles/r04/decoder_persistent_lanes_ab.txt).
            // Knob decoder_persist: 0 automatic, 1 always the persistent form, 2 never.
            const int persist = knob(kKnobDecoderPersist);
            if (cfg == kLane4Config && persist != 2) {
                constexpr int R_ = kLane4Config % 1000, P_ = (kLane4Config / 1000 & 2) ? 32 : 64, FU_ = (kLane4Config / 1000 & 1) ? 128 : 64,
                              FS_ = (kLane4Config / 1000 & 4) ? 1 : 2, FE_ = (kLane4Config / 1000 & 8) ? 2 : 1, IE_ = (kLane4Config / 1000 & 16) ? 2 : 1,
                              POL_ = (kLane4Config / 1000 & 32) ? 16 : 0;
                static std::atomic<int> per_cu_cached[64];
                int per_cu = dev >= 0 && dev < 64 ? per_cu_cached[dev].load(std::memory_order_relaxed) : 0;
                if (per_cu <= 0) {
                    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_lane4_persistent_kernel<true, R_, P_, FU_, FS_, FE_, IE_, POL_>, 64, 0));
                    if (per_cu <= 0) per_cu = 1;
                    if (dev >= 0 && dev < 64) per_cu_cached[dev].store(per_cu, std::memory_order_relaxed);
                }
                const int64_t capacity = (int64_t)per_cu * cus;
                // 0: one block per lane only; 1: persistent only; 2: counted, one of the two
                // (a batch that fits the residency in one round gives every lane one block either way: nothing to refill, no counter needed)
                int mode = persist == 1 ? 1 : ((int64_t)grid <= capacity ? 0 : ((int64_t)grid < 3 * capacity ? 1 : (lane_filter == kAllBlocks ? 0 : 2)));
                unsigned long long* counter = nullptr;
                CounterLease counter_lease;                              // (its destructor records the slot's event behind the kernels queued below)
                if (mode != 0) { int rc = decoder_counter(dev, stream, &counter, counter_lease); if (rc) return rc; }
                const unsigned* gate = mode == 2 ? (const unsigned*)(counter + 1) : nullptr;       // (the slot's second qword: the count)
                const unsigned threshold = (unsigned)(d.n_blocks - d.n_blocks / 10);            // "nearly every block": 90 %
                if (mode == 2) {
                    const unsigned cg = (unsigned)((d.n_blocks + 255) / 256 < 512 ? (d.n_blocks + 255) / 256 : 512);   // (one atomic per wavefront on ONE address: few, fat wavefronts)
                    hipLaunchKernelGGL(count_selected_kernel, dim3(cg), dim3(256), 0, stream, d, lane_filter, (unsigned*)(counter + 1));
                    HIP_TRY(hipGetLastError());
                }
                unsigned pg = (unsigned)(capacity < (int64_t)grid ? capacity : (int64_t)grid);
                if (knob(kKnobDecoderGroups) > 0 && (unsigned)knob(kKnobDecoderGroups) < pg) pg = (unsigned)knob(kKnobDecoderGroups);
                // one residency round at most, and few wavefronts per CU: workgroups of four wavefronts, so that the SIMDs of a CU share them evenly
                // whatever ran before (decode_lane4_wg4_kernel)
                // (from more than one wavefront per CU on: with at most one, single-wavefront workgroups spread evenly by themselves and keep the
                //  dual ring stores, 8.19 vs 8.59 ms at 16 384 D2 blocks; knob decoder_wg4 = 2 takes the four-wavefront form from four wavefronts on: tests)
                const int wg4_knob = knob(kKnobDecoderWg4);
                const bool wg4 = mode == 0 && wg4_knob != 1 && (int64_t)grid >= 4 && (wg4_knob == 3 || ((int64_t)grid > (wg4_knob == 2 ? 0 : (int64_t)cus) && (int64_t)grid <= capacity));   // (up to one residency round; 3: whatever the batch size -- A/B runs)
                auto both_forms = [&](auto pol_tag) -> int {
                    constexpr int POLX = decltype(pol_tag)::value;
                    if (wg4) {
                        const unsigned g4 = (grid + 3u) / 4u;
                        if (known) hipLaunchKernelGGL((decode_lane4_wg4_kernel<true, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(g4), dim3(256), 0, stream, d, lane_filter);
                        else       hipLaunchKernelGGL((decode_lane4_wg4_kernel<false, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(g4), dim3(256), 0, stream, d, lane_filter);
                        HIP_TRY(hipGetLastError());
                        return 0;
                    }
                    if (mode != 1) {
                        if (known) hipLaunchKernelGGL((decode_lane4_kernel<true, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(grid), dim3(64), 0, stream, d, lane_filter, gate, mode == 2 ? 1 : 0, threshold);
                        else       hipLaunchKernelGGL((decode_lane4_kernel<false, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(grid), dim3(64), 0, stream, d, lane_filter, gate, mode == 2 ? 1 : 0, threshold);
                        HIP_TRY(hipGetLastError());
                    }
                    if (mode != 0) {
                        if (known) hipLaunchKernelGGL((decode_lane4_persistent_kernel<true, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(pg), dim3(64), 0, stream, d, lane_filter, counter, gate, mode == 2 ? 2 : 0, threshold);
                        else       hipLaunchKernelGGL((decode_lane4_persistent_kernel<false, R_, P_, FU_, FS_, FE_, IE_, POLX>), dim3(pg), dim3(64), 0, stream, d, lane_filter, counter, gate, mode == 2 ? 2 : 0, threshold);
                    }
                    return 0;
                };
                const int rc2 = wrapped ? both_forms(std::integral_constant<int, POL_ | 32>{}) : both_forms(std::integral_constant<int, POL_>{});
                if (rc2) return rc2;
            } else
            switch (cfg) {
            LZ4HIP_LANE4_CASE(kLane4Config);
#ifdef LZ4HIP_TUNING_BUILD                                              /* residency / ring / piece / flush-unit sweeps (tools/ab_decoder_knobs.py) */
            LZ4HIP_LANE4_CASE(128); LZ4HIP_LANE4_CASE(2128); LZ4HIP_LANE4_CASE(6128); LZ4HIP_LANE4_CASE(3192); LZ4HIP_LANE4_CASE(192); LZ4HIP_LANE4_CASE(1192); LZ4HIP_LANE4_CASE(2192);
            LZ4HIP_LANE4_CASE(5192); LZ4HIP_LANE4_CASE(7192); LZ4HIP_LANE4_CASE(11192); LZ4HIP_LANE4_CASE(25192); LZ4HIP_LANE4_CASE(11256); LZ4HIP_LANE4_CASE(1256); LZ4HIP_LANE4_CASE(3256); LZ4HIP_LANE4_CASE(7256); LZ4HIP_LANE4_CASE(5256);
            LZ4HIP_LANE4_CASE(27192); LZ4HIP_LANE4_CASE(43192); LZ4HIP_LANE4_CASE(35192); LZ4HIP_LANE4_CASE(58128); LZ4HIP_LANE4_CASE(59256); LZ4HIP_LANE4_CASE(59224); LZ4HIP_LANE4_CASE(59208);
            LZ4HIP_LANE4_CASE(59384); LZ4HIP_LANE4_CASE(59512); LZ4HIP_LANE4_CASE(59768); LZ4HIP_LANE4_CASE(59960);   /* round 6: large rings for batches that leave the LDS idle anyway (tools/r06/call12.sh) */
#endif
            default: return fail(LZ4HIP_E_ARGUMENT, "decoder_ring: this library has no generation-4 lane decoder with that configuration");
            }
#undef LZ4HIP_LANE4_CASE
#undef LZ4HIP_LAUNCH_LANE4
#undef LZ4HIP_LAUNCH_LANE4_POL
        }
#ifdef LZ4HIP_TUNING_BUILD                                              /* the generations it replaced (tools/ab/), for same-box A/B runs */
        else if (gen == 2) {
            if (known) hipLaunchKernelGGL((decode_lane_kernel<true, 128, 64>), dim3(grid), dim3(64), 0, stream, d, lane_filter);
            else       hipLaunchKernelGGL((decode_lane_kernel<false, 128, 64>), dim3(grid), dim3(64), 0, stream, d, lane_filter);
        } else if (gen == 3) {
            const int ring = knob(kKnobDecoderRing) ? knob(kKnobDecoderRing) : 128;
#define LZ4HIP_LAUNCH_LANE3(RING)                                                                                               \
            do {                                                                                                                \
                if (known) hipLaunchKernelGGL((decode_lane3_kernel<true, RING, 64>), dim3(grid), dim3(64), 0, stream, d, lane_filter);  \
                else       hipLaunchKernelGGL((decode_lane3_kernel<false, RING, 64>), dim3(grid), dim3(64), 0, stream, d, lane_filter); \
            } while (0)
            switch (ring) {
            case 128: LZ4HIP_LAUNCH_LANE3(128); break;
            case 192: LZ4HIP_LAUNCH_LANE3(192); break;
            case 240: LZ4HIP_LAUNCH_LANE3(240); break;
            default: return fail(LZ4HIP_E_ARGUMENT, "decoder_ring: this library has no generation-3 lane decoder with that ring size");
            }
#undef LZ4HIP_LAUNCH_LANE3
        }
#endif
        else return fail(LZ4HIP_E_ARGUMENT, "decoder_gen: this library has no lane decoder of that generation");
        count_dispatch(LZ4HIP_K_DECODE_LANE);
    }
    if (wave_filter >= 0) {
        const unsigned waves = kWaveDecodeWavesPerGroup, grid = (unsigned)((d.n_blocks + waves - 1) / waves);
        if (known) hipLaunchKernelGGL(decode_kernel<true>, dim3(grid), dim3(64 * waves), 0, stream, d, wave_filter);
        else       hipLaunchKernelGGL(decode_kernel<false>, dim3(grid), dim3(64 * waves), 0, stream, d, wave_filter);
        count_dispatch(LZ4HIP_K_DECODE_WAVE);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Pinned (page-locked) host staging, grow-only, per calling thread: two slots so that the host can fill / drain
// one slice while the device works on the other.
struct Pinned {
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = n < (1u << 20) ? (1u << 20) : n;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
    ~Pinned() { /* see ~Scratch */ }
};

// Per calling thread and device: the streams and events of the host pipeline (H2D | kernels | D2H).  A slice owns one of
// kHostSlots slots (pinned in/out images, device in/out images, a kernel stream): the kernels of consecutive slices run
// on DIFFERENT streams, so that a slice's kernels start as soon as its input has landed, next to those of the slices
// before it -- a launch over a thousand blocks cannot fill the device and takes the same few milliseconds as one over
// four thousand.
constexpr int kHostSlots = 4;
struct HostPipe {
    bool ready = false;
    int dev = -1;                // streams and events belong to a device: one set per (thread, device)
    hipStream_t s_in = nullptr, s_out = nullptr, s_k[kHostSlots];
    hipEvent_t e_in[kHostSlots], e_k[kHostSlots], e_out[kHostSlots], e_start;
    int init()
    {
        if (ready) return 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e_start, hipEventDisableTiming));
        for (int k = 0; k < kHostSlots; k++) {
            HIP_TRY(hipStreamCreateWithFlags(&s_k[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&e_in[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e_k[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e_out[k], hipEventDisableTiming));
        }
        ready = true;
        return 0;
    }
};
// Everything a calling thread needs to stage host batches through ONE device: scratch, pinned slots, pipeline.
// Indexed by the device that is current when the call is made, so a thread may hipSetDevice() between calls
// (and the multi-device entry points run one worker thread per device).
struct HostContext {
    Scratch scratch;
    Pinned pin_in[kHostSlots], pin_out[kHostSlots];
    HostPipe pipe;
};
// One per (thread, device), created on first use and kept for the life of the thread.  The threads that call this are
// the callers' own (single-device entry points) and the library's PERSISTENT device workers (multi-device entry points,
// below) -- never a thread the library starts per call.
HostContext* host_context(int dev, bool create = true)
{
    static thread_local HostContext* ctx[64] = {};
    if (dev < 0 || dev >= 64) return nullptr;
    if (!ctx[dev] && create) ctx[dev] = new HostContext();   // lives as long as the thread (see ~Scratch)
    return ctx[dev];
}
void release_host_context(int dev)
{
    if (HostContext* hc = host_context(dev, false)) {
        hc->scratch.release();
        for (int k = 0; k < kHostSlots; k++) { hc->pin_in[k].release(); hc->pin_out[k].release(); }
    }
}

// ---- persistent device workers of the multi-device entry points --------------------------------------------------------
// One thread per (logical) device, started on first use and kept until the process ends: its staging context (device
// images, pinned slots, streams, events) is allocated once and reused by every later call.  A caller posts one job per
// worker and waits for all of them; concurrent callers queue on the worker's `busy` mutex (taken in ascending worker order).
struct DeviceWorker {
    std::mutex busy;                 // held by the caller that owns the worker for one job
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, finished = false, started = false;
    void loop()
    {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job; });
                j = std::move(job); has_job = false;
            }
            try { j(); } catch (...) {}                              // (jobs report through their own state; see run_host_batch_multi)
            {
                std::lock_guard<std::mutex> lk(mu);
                finished = true;
            }
            cv.notify_all();
        }
    }
    // (caller holds `busy`)
    int post(std::function<void()> j)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            try { std::thread(&DeviceWorker::loop, this).detach(); }
            catch (const std::system_error& e) { return fail(LZ4HIP_E_MEMORY, std::string("cannot start a device worker thread: ") + e.what()); }
            started = true;
        }
        job = std::move(j); has_job = true; finished = false;
        lk.unlock();
        cv.notify_all();
        return 0;
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished; });
    }
};
constexpr int kMaxWorkers = 64;
DeviceWorker* g_worker[kMaxWorkers];         // never freed: a worker blocked in cv.wait at process exit must keep its object
std::mutex g_worker_mu;
DeviceWorker* device_worker(int logical)
{
    std::lock_guard<std::mutex> lk(g_worker_mu);
    if (!g_worker[logical]) g_worker[logical] = new DeviceWorker();
    return g_worker[logical];
}

// ---- the row pool: gathers and scatters between caller memory and the pinned staging ------------------------------------
// Plain memcpy of rows, ~10 GB/s per core -- and with six slices per batch the calling thread used to spend more time in them
// than PCIe needs for the payload (round 3: 16 threads, started and joined per call: 24 GB/s for a 16 384-block decode).  Now
// ONE persistent pool per process, started on first use: min(hardware threads / 4, 64) threads (knob host_threads), shared by
// every caller -- the device workers of a multi-device call included: each of their jobs is cut into chunks that any pool
// thread (and the submitting thread) may take, so eight devices share the pool instead of getting two threads each.
struct RowJob {
    std::function<void(int64_t)> f;
    int64_t n = 0, chunk = 1;
    std::atomic<int64_t> next{ 0 }, remaining{ 0 };
    std::mutex m;
    std::condition_variable done_cv;
    bool exhausted() const { return next.load(std::memory_order_relaxed) >= n; }
    // runs one chunk; false when none was left
    bool work_one()
    {
        const int64_t lo = next.fetch_add(chunk, std::memory_order_relaxed);
        if (lo >= n) return false;
        const int64_t hi = lo + chunk < n ? lo + chunk : n;
        for (int64_t i = lo; i < hi; i++) f(i);
        if (remaining.fetch_sub(hi - lo, std::memory_order_acq_rel) == hi - lo) {
            std::lock_guard<std::mutex> lk(m);
            done_cv.notify_all();
        }
        return true;
    }
};
struct RowPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<RowJob>> q;                            // in priority order: gathers (they feed the pipeline) before scatters
    unsigned started = 0;
    void loop()
    {
        for (;;) {
            std::shared_ptr<RowJob> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    while (!q.empty() && q.front()->exhausted()) q.pop_front();
                    for (auto& c : q) if (!c->exhausted()) { j = c; break; }
                    if (j) break;
                    cv.wait(lk);
                }
            }
            j->work_one();                                            // one chunk, then look again: a more urgent job may have arrived
        }
    }
    unsigned want_threads()
    {
        if (knob(kKnobHostThreads) > 0) return (unsigned)knob(kKnobHostThreads);
        // a quarter of the hardware threads, at least 8 where the host has them, at most 64 -- and never more than the host has
        const unsigned hc = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1u;
        unsigned t = hc / 4;
        t = t < 8 ? 8 : (t > 64 ? 64 : t);
        return t > hc ? hc : t;
    }
    // queues the job and returns at once; wait() (which also works on it) before anything it touches is reused
    std::shared_ptr<RowJob> submit(int64_t n, std::function<void(int64_t)> f, bool urgent)
    {
        const unsigned want = want_threads();
        auto j = std::make_shared<RowJob>();
        j->f = std::move(f); j->n = n;
        j->chunk = n / ((int64_t)want * 4) > 0 ? n / ((int64_t)want * 4) : 1;
        j->remaining.store(n);
        {
            std::lock_guard<std::mutex> lk(mu);
            while (started + 1 < want) {                              // (the submitting thread works too)
                try { std::thread(&RowPool::loop, this).detach(); }
                catch (const std::system_error&) { break; }           // no more threads to be had: fewer helpers, same result
                started++;
            }
            if (urgent) q.push_front(j); else q.push_back(j);
        }
        cv.notify_all();
        return j;
    }
    void wait(const std::shared_ptr<RowJob>& j)
    {
        while (j->work_one()) {}
        std::unique_lock<std::mutex> lk(j->m);
        j->done_cv.wait(lk, [&] { return j->remaining.load(std::memory_order_acquire) == 0; });
    }
    void run(int64_t n, std::function<void(int64_t)> f) { wait(submit(n, std::move(f), true)); }
};
RowPool* row_pool()
{
    static RowPool* p = new RowPool();                               // never destroyed: its threads outlive main()
    return p;
}

// f(i) for i in [0, n): on the calling thread for small jobs, on the row pool for large ones.
template <class F>
void for_rows(int64_t n, size_t bytes, F f)
{
    if (bytes < (4u << 20) || n < 2) { for (int64_t i = 0; i < n; i++) f(i); return; }
    try { row_pool()->run(n, std::function<void(int64_t)>(f)); }
    catch (const std::bad_alloc&) { for (int64_t i = 0; i < n; i++) f(i); }   // (idempotent copies: doing them again is harmless)
}

// Stage a host batch through device memory, run `run` on it, copy results (and dst payloads) back.
// The batch is cut into slices; per slice: rows are gathered into pinned memory (host threads), ONE host-to-device copy,
// the kernels, ONE device-to-host copy into pinned memory, rows scattered to the caller.  Copies in, kernels and copies
// out run on their own streams over kHostSlots sets of buffers: slice k+1 travels in and slice k-1 out while slice k is
// being processed (PCIe is full duplex), kernels of neighbouring slices overlap, and the host gathers / scatters next to
// all that.
template <class Run>
// slice_hint > 0: blocks per slice wanted by the caller (LZ4HC: its lane kernels need ~0.35 s however few blocks a launch has, so
// a batch goes in slices of up to 16384 blocks instead of ~2048).
int run_host_batch(const lz4hip_batch_t* hb, bool dst_len_is_result, Run run, int64_t slice_hint = 0)
{
    int rc = check_batch(hb);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    const int64_t n = hb->n_blocks;
    if (n == 0) return 0;

    // tight device layout: per-block slots of the maximum length, 16-byte aligned
    int64_t max_src = 0, max_dst = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t sl = hb->src_len ? hb->src_len[i] : hb->src_len_all;
        const int32_t dc = hb->dst_cap ? hb->dst_cap[i] : hb->dst_cap_all;
        if (sl < 0) return fail(LZ4HIP_E_ARGUMENT, "negative source length");
        max_src = sl > max_src ? sl : max_src;
        max_dst = dc > max_dst ? dc : max_dst;
    }
    const size_t s_stride = align_up((size_t)max_src + 16, 16), d_stride = align_up((size_t)max_dst + 16, 16);
    // slice size: a small batch goes in one piece; a large one in about 6 slices (knob host_slices; profiles/r02/host_slices_sweep.txt) of
    // 32 MiB .. 512 MiB of rows each
    const size_t row_bytes = s_stride + d_stride;
    // slices: the kernels of a slice take milliseconds however few blocks it has (a 64 KiB block of short sequences needs 2.6 ms
    // in the wavefront mapping), so small batches are cut less: 1 slice up to ~3 k blocks, 2 at 4 k, 6 from 12 k
    // (profiles/r03/host_slices_by_batch_size.txt: 4 096 blocks 19.4 GB/s with 2 slices, 14.7-18.4 with 6-8)
    const int auto_slices = (int)(n / 2048 < 1 ? 1 : (n / 2048 > 6 ? 6 : n / 2048));
    const int want_slices = knob(kKnobHostSlices) > 0 ? knob(kKnobHostSlices) : auto_slices;
    int64_t per_slice = (n + want_slices - 1) / want_slices;
    int64_t lo = (int64_t)((32u << 20) / row_bytes), hi = (int64_t)((512u << 20) / row_bytes);
    if (slice_hint > 0 && knob(kKnobHostSlices) <= 0 && max_src <= 65536) {
        // (the hint is for the LZ4HC kernels of blocks <= 64 KiB; whatever the rows are, a slice stays below 4 GiB of staging)
        const int64_t cap4g = (int64_t)((4ull << 30) / row_bytes);
        const int64_t hinted = slice_hint < (hi > cap4g ? hi : cap4g) ? slice_hint : (hi > cap4g ? hi : cap4g);
        per_slice = hinted; hi = hi > hinted ? hi : hinted;
    }
    per_slice = per_slice < lo ? lo : per_slice;
    per_slice = per_slice > hi ? hi : per_slice;
    per_slice = per_slice < 1 ? 1 : (per_slice > n ? n : per_slice);
    const size_t m = (size_t)per_slice;
    // The slice table (equal slices; a quarter slice at both ends -- less time before the first kernel and after the last one -- measured
    // no gain: profiles/r06/host_tapered_slices_ab.txt, tools/ab/host_tapered_slices.patch).
    std::vector<int64_t> bounds;
    try {
        for (int64_t at = 0; at < n; at += per_slice) bounds.push_back(at);
        bounds.push_back(n);
    } catch (const std::bad_alloc&) { return fail(LZ4HIP_E_MEMORY, "out of host memory"); }
    const int64_t n_slices = (int64_t)bounds.size() - 1;
    const int slots = n_slices < kHostSlots ? (int)n_slices : kHostSlots;
    // device and pinned "in" image: [src slots | src_len | dst_cap];  "out" image: [dst slots | result]
    const size_t in_lens = align_up(s_stride * m, 256), in_caps = in_lens + align_up(4 * m, 256), in_bytes = in_caps + align_up(4 * m, 256);
    const size_t out_res = align_up(d_stride * m, 256), out_bytes = out_res + align_up(4 * m, 256);
    int dev_now = 0;
    HIP_TRY(hipGetDevice(&dev_now));
    HostContext* hc = host_context(dev_now);
    if (!hc) return fail(LZ4HIP_E_DEVICE, "device index out of range");
    Scratch& g_scratch = hc->scratch;
    Pinned* g_pin_in = hc->pin_in;
    Pinned* g_pin_out = hc->pin_out;
    if ((rc = g_scratch.reserve((size_t)slots * (in_bytes + out_bytes)))) return rc;
    for (int k = 0; k < slots; k++) {
        if ((rc = g_pin_in[k].reserve(in_bytes))) return rc;
        if ((rc = g_pin_out[k].reserve(out_bytes))) return rc;
    }
    if ((rc = hc->pipe.init())) return rc;
    HostPipe& pp = hc->pipe;
    uint8_t* d_in[kHostSlots];
    uint8_t* d_out[kHostSlots];
    for (int k = 0; k < slots; k++) {
        d_in[k] = (uint8_t*)g_scratch.p + (size_t)k * (in_bytes + out_bytes);
        d_out[k] = d_in[k] + in_bytes;
    }

    auto src_row = [&](int64_t i) { return (const uint8_t*)hb->src + (hb->src_off ? hb->src_off[i] : i * hb->src_stride); };
    auto src_len = [&](int64_t i) { return hb->src_len ? hb->src_len[i] : hb->src_len_all; };
    auto dst_cap = [&](int64_t i) { return hb->dst_cap ? hb->dst_cap[i] : hb->dst_cap_all; };

    // drain slice [first, first + cnt) from pinned slot `slot` into the caller's buffers
    // The scatter of a slice runs on the row pool WITHOUT the calling thread waiting for it: that thread goes on gathering
    // the next slice (round 3 did gather k+1 and scatter k-1 one after the other: ~23 of the 34 ms of a 16 384-block decode).
    // A slot's pinned buffer is reused only after its scatter has finished.
    std::shared_ptr<RowJob> scatter_job[kHostSlots];
    auto scatter_wait = [&](int slot) {
        if (scatter_job[slot]) { row_pool()->wait(scatter_job[slot]); scatter_job[slot].reset(); }
    };
    auto scatter = [&](int64_t first, int64_t cnt, int slot) {
        const uint8_t* po = (const uint8_t*)g_pin_out[slot].p;
        const int32_t* res = (const int32_t*)(po + out_res);
        for (int64_t j = 0; j < cnt; j++) hb->result[first + j] = res[j];
        auto row = [=](int64_t j) {
            // bytes the caller gets back: the result for encoders / unknown-size decode, the full size for known-size decode
            const int32_t cap = hb->dst_cap ? hb->dst_cap[first + j] : hb->dst_cap_all;
            int64_t nbytes = dst_len_is_result ? res[j] : cap;
            if (!dst_len_is_result && res[j] < 0) nbytes = 0;
            if (nbytes > cap) nbytes = cap;
            uint8_t* to = (uint8_t*)hb->dst + (hb->dst_off ? hb->dst_off[first + j] : (first + j) * hb->dst_stride);
            if (nbytes > 0) memcpy(to, po + d_stride * (size_t)j, (size_t)nbytes);
        };
        if ((size_t)cnt * d_stride < (4u << 20) || cnt < 2) { for (int64_t j = 0; j < cnt; j++) row(j); return; }
        try { scatter_job[slot] = row_pool()->submit(cnt, std::function<void(int64_t)>(row), false); }
        catch (const std::bad_alloc&) { for (int64_t j = 0; j < cnt; j++) row(j); }
    };
#define PIPE_TRY(expr) do { if ((expr) != hipSuccess) { err = fail(LZ4HIP_E_DEVICE, #expr " failed"); } } while (0)

    int err = 0;
    int64_t drained = 0;                                             // slices [0, drained) are back in the caller's buffers
    auto drain_one = [&]() {                                         // (blocking)
        const int slot = (int)(drained % slots);
        PIPE_TRY(hipEventSynchronize(pp.e_out[slot]));
        if (!err) scatter(bounds[(size_t)drained], bounds[(size_t)drained + 1] - bounds[(size_t)drained], slot);
        drained++;
    };
    // whatever the caller queued on its stream before this call comes first
    PIPE_TRY(hipEventRecord(pp.e_start, hipStreamPerThread));
    PIPE_TRY(hipStreamWaitEvent(pp.s_in, pp.e_start, 0));
    int64_t slice = 0;
    for (; slice < n_slices && !err; slice++) {
        const int64_t first = bounds[(size_t)slice], cnt = bounds[(size_t)slice + 1] - first;
        const int slot = (int)(slice % slots);
        while (!err && drained + slots <= slice) drain_one();        // the slot's previous slice must be out of its buffers
        scatter_wait(slot);                                          // ... and in the caller's
        if (err) break;
        uint8_t* pi = (uint8_t*)g_pin_in[slot].p;
        int32_t* lens = (int32_t*)(pi + in_lens);
        int32_t* caps = (int32_t*)(pi + in_caps);
        for (int64_t j = 0; j < cnt; j++) { lens[j] = src_len(first + j); caps[j] = dst_cap(first + j); }
        for_rows(cnt, (size_t)cnt * s_stride, [&, pi, first](int64_t j) {
            const int32_t sl = src_len(first + j);
            if (sl > 0) memcpy(pi + s_stride * (size_t)j, src_row(first + j), (size_t)sl);
        });
        // copy in (the slot's previous kernels have finished: their slice has been drained)
        // (the rows the slice has, then its lengths and capacities: a short slice does not pay for a full one's image)
        PIPE_TRY(hipMemcpyAsync(d_in[slot], pi, s_stride * (size_t)cnt, hipMemcpyHostToDevice, pp.s_in));
        PIPE_TRY(hipMemcpyAsync(d_in[slot] + in_lens, pi + in_lens, in_bytes - in_lens, hipMemcpyHostToDevice, pp.s_in));
        PIPE_TRY(hipEventRecord(pp.e_in[slot], pp.s_in));
        // kernels, on the slot's own stream: after their input has landed
        hipStream_t ks = pp.s_k[slot];
        PIPE_TRY(hipStreamWaitEvent(ks, pp.e_in[slot], 0));
        if (err) break;
        lz4hip_batch_t db;
        db.src = d_in[slot]; db.src_off = nullptr; db.src_stride = (int64_t)s_stride; db.src_len = (const int32_t*)(d_in[slot] + in_lens);
        db.dst = d_out[slot]; db.dst_off = nullptr; db.dst_stride = (int64_t)d_stride; db.dst_cap = (const int32_t*)(d_in[slot] + in_caps);
        db.dst_cap_all = 0; db.src_len_all = (int32_t)max_src;   /* upper-bound hint */ db.result = (int32_t*)(d_out[slot] + out_res); db.n_blocks = cnt;
        if ((err = run(&db, ks))) break;
        PIPE_TRY(hipEventRecord(pp.e_k[slot], ks));
        // copy out: after the kernels
        PIPE_TRY(hipStreamWaitEvent(pp.s_out, pp.e_k[slot], 0));
        PIPE_TRY(hipMemcpyAsync(g_pin_out[slot].p, d_out[slot], d_stride * (size_t)cnt, hipMemcpyDeviceToHost, pp.s_out));
        PIPE_TRY(hipMemcpyAsync((uint8_t*)g_pin_out[slot].p + out_res, d_out[slot] + out_res, out_bytes - out_res, hipMemcpyDeviceToHost, pp.s_out));
        PIPE_TRY(hipEventRecord(pp.e_out[slot], pp.s_out));
        if (err) break;
        // slices that have already arrived are scattered while the later ones are in flight
        while (!err && drained < slice && hipEventQuery(pp.e_out[drained % slots]) == hipSuccess) drain_one();
    }
    while (!err && drained < slice) drain_one();
    for (int k = 0; k < kHostSlots; k++) scatter_wait(k);           // (also on error: the jobs read this frame's buffers)
#undef PIPE_TRY
    if (err) {
        (void)hipStreamSynchronize(pp.s_in); (void)hipStreamSynchronize(pp.s_out);
        for (int k = 0; k < kHostSlots; k++) (void)hipStreamSynchronize(pp.s_k[k]);
    }
    return err;
}

// Host-resident batch sharded over the devices of `device_mask` (bit d = HIP device d; 0 = every visible device):
// block i belongs to the (i mod N)-th selected device -- the partition of SURVEY.md 8e / BASELINE configs[4] -- one
// persistent worker thread per device, each with its own staging pipeline (run_host_batch on that device); no device ever
// sees another device's blocks and nothing is exchanged between them.  Results land in the caller's arrays in global order.
// (The "logical_devices" knob makes N workers out of fewer devices, wrapping around: how the threaded path is tested on
// a one-GPU box.)
template <class Run>
int run_host_batch_multi(const lz4hip_batch_t* hb, bool dst_len_is_result, uint64_t device_mask, Run run, int64_t slice_hint = 0, int workers_on_one_device = 0)
{
    int rc = check_batch(hb);
    if (rc) return rc;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return fail(LZ4HIP_E_DEVICE, "no HIP device");
    std::vector<int> devs;
    for (int d = 0; d < visible && d < 64; d++)
        if (device_mask == 0 || ((device_mask >> d) & 1ull)) devs.push_back(d);
    if (devs.empty()) return fail(LZ4HIP_E_ARGUMENT, "device_mask selects no visible device");
    const int logical = workers_on_one_device > 0 ? workers_on_one_device : knob(kKnobLogicalDevices);
    if (logical > 0) {
        const std::vector<int> base = devs;
        devs.clear();
        for (int k = 0; k < logical && k < kMaxWorkers; k++) devs.push_back(base[(size_t)k % base.size()]);
    }
    const int64_t n = hb->n_blocks;
    if (n == 0) return 0;
    const int nd = (int)devs.size();

    struct Shard {
        std::vector<int64_t> src_off, dst_off;
        std::vector<int32_t> src_len, dst_cap, result;
        lz4hip_batch_t b;
        int rc = 0;
        std::string err;
    };
    std::vector<Shard> shards((size_t)nd);
    for (int k = 0; k < nd; k++) {
        Shard& sh = shards[(size_t)k];
        const int64_t cnt = n > k ? (n - k + nd - 1) / nd : 0;
        sh.src_off.resize((size_t)cnt); sh.dst_off.resize((size_t)cnt);
        sh.src_len.resize((size_t)cnt); sh.dst_cap.resize((size_t)cnt); sh.result.assign((size_t)cnt, 0);
        for (int64_t j = 0; j < cnt; j++) {
            const int64_t i = j * nd + k;
            sh.src_off[(size_t)j] = hb->src_off ? hb->src_off[i] : i * hb->src_stride;
            sh.dst_off[(size_t)j] = hb->dst_off ? hb->dst_off[i] : i * hb->dst_stride;
            sh.src_len[(size_t)j] = hb->src_len ? hb->src_len[i] : hb->src_len_all;
            sh.dst_cap[(size_t)j] = hb->dst_cap ? hb->dst_cap[i] : hb->dst_cap_all;
        }
        sh.b = *hb;
        sh.b.src_off = sh.src_off.data(); sh.b.dst_off = sh.dst_off.data();
        sh.b.src_len = sh.src_len.data(); sh.b.dst_cap = sh.dst_cap.data();
        sh.b.result = sh.result.data(); sh.b.n_blocks = cnt;
    }
    if (nd == 1) {                                                   // one device: on the calling thread, with its own context
        int prev_dev = 0;
        HIP_TRY(hipGetDevice(&prev_dev));
        HIP_TRY(hipSetDevice(devs[0]));
        rc = run_host_batch(&shards[0].b, dst_len_is_result, run, slice_hint);
        (void)hipSetDevice(prev_dev);
        if (rc) return rc;
    } else {
        std::vector<DeviceWorker*> workers((size_t)nd);
        for (int k = 0; k < nd; k++) workers[(size_t)k] = device_worker(k);
        int posted = 0, post_rc = 0;
        for (int k = 0; k < nd; k++) {                               // ascending order: concurrent callers cannot deadlock
            DeviceWorker* w = workers[(size_t)k];
            w->busy.lock();
            Shard* sh = &shards[(size_t)k];
            const int phys = devs[(size_t)k];
            const unsigned share = (unsigned)nd;
            post_rc = w->post([sh, phys, share, dst_len_is_result, run, slice_hint] {
                if (sh->b.n_blocks == 0) return;
                if (hipSetDevice(phys) != hipSuccess) { sh->rc = LZ4HIP_E_DEVICE; sh->err = "hipSetDevice failed"; return; }
                (void)share;
                // nothing may be thrown across the worker's loop (a detached thread: std::terminate for the whole process)
                try { sh->rc = run_host_batch(&sh->b, dst_len_is_result, run, slice_hint); }
                catch (const std::bad_alloc&) { sh->rc = LZ4HIP_E_MEMORY; g_last_error = "out of host memory in a device worker"; }
                catch (const std::exception& e) { sh->rc = LZ4HIP_E_DEVICE; g_last_error = std::string("device worker: ") + e.what(); }
                if (sh->rc) sh->err = g_last_error;                  // thread-local: carry it back to the caller
            });
            if (post_rc) { w->busy.unlock(); break; }
            posted++;
        }
        for (int k = 0; k < posted; k++) { workers[(size_t)k]->wait(); workers[(size_t)k]->busy.unlock(); }
        if (post_rc) return post_rc;
    }
    for (int k = 0; k < nd; k++) {
        const Shard& sh = shards[(size_t)k];
        if (sh.rc) return fail(sh.rc, "device " + std::to_string(devs[(size_t)k]) + ": " + sh.err);
        for (int64_t j = 0; j < sh.b.n_blocks; j++) hb->result[j * nd + k] = sh.result[(size_t)j];
    }
    return 0;
}

int single(const char* src, int src_len, char* dst, int dst_cap, int kind /*0 fast,1 hc,2 dec known,3 dec unknown*/)
{
    if (!src || !dst) return fail(LZ4HIP_E_ARGUMENT, "NULL buffer");
    if (src_len < 0) return fail(LZ4HIP_E_ARGUMENT, "negative length");
    int32_t result = 0;
    lz4hip_batch_t b;
    b.src = src; b.src_off = nullptr; b.src_stride = 0; b.src_len = nullptr; b.src_len_all = src_len;
    b.dst = dst; b.dst_off = nullptr; b.dst_stride = 0; b.dst_cap = nullptr; b.dst_cap_all = dst_cap < 0 ? 0 : dst_cap;
    b.result = &result; b.n_blocks = 1;
    int rc = kind <= 1 ? lz4hip_encode_batch_host(&b, kind) : lz4hip_decode_batch_host(&b, kind == 2);
    return rc ? rc : result;
}

// Number of source bytes a known-size decode of `osize` output bytes will look at.  This only walks
// the token / length bytes (no byte of payload is decoded on the host): LZ4_uncompress() is not
// told its input size (original/lz4.c:812-814), so the extent of the H2D copy has to come from the
// stream itself.  Mirrors the control flow of lz4hip::decode_block<true>.
int known_size_extent(const uint8_t* src, int osize)
{
    int64_t ip = 0, op = 0;
    for (;;) {
        unsigned token = src[ip++];
        int64_t ll = token >> 4;
        if (ll == 15) { unsigned b; do { b = src[ip++]; ll += b; } while (b == 255 && ll < (1 << 30)); }
        if (op + ll > (int64_t)osize - 8) return (int)(op + ll == osize ? ip + ll : ip);
        ip += ll; op += ll;
        ip += 2;
        int64_t ml = token & 15;
        if (ml == 15) { while (src[ip] == 255 && ml < (1 << 30)) { ml += 255; ip++; } ml += src[ip++]; }
        op += ml + 4;
        if (op > (int64_t)osize - 5) return (int)ip;
    }
}

}  // namespace

extern "C" {

const char* lz4hip_last_error(void) { return g_last_error.c_str(); }

int lz4hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* lz4hip_codec_name(void)
{
    static thread_local std::string name;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        name = "HIP gfx950 (no device)";
    } else {
        name = std::string("HIP ") + prop.gcnArchName + " (" + prop.name + ")";
    }
    return name.c_str();
}

int lz4hip_compressBound(int isize) { return isize + isize / 255 + 16; }

// Which sources this binary was built from: lz4net_amd/build.py hashes csrc/ (the hash bench.py and the committed counter files are keyed on) and
// passes it in; the marker string is also how build.py recognises a stale prebuilt library without loading it.
#ifndef LZ4HIP_CSRC_SHA
#define LZ4HIP_CSRC_SHA "unknown"
#endif
#ifdef LZ4HIP_TUNING_BUILD
#define LZ4HIP_BUILD_KIND "+tuning"
#else
#define LZ4HIP_BUILD_KIND ""
#endif
const char* lz4hip_build_id(void)
{
    static const char marker[] = "LZ4HIP_BUILD_ID=" LZ4HIP_CSRC_SHA LZ4HIP_BUILD_KIND;
    return marker + 16;
}

int lz4hip_dispatch_counts(uint64_t* counts, int n)
{
    for (int k = 0; counts && k < n && k < LZ4HIP_K_COUNT; k++) counts[k] = g_dispatch[k].load(std::memory_order_relaxed);
    return LZ4HIP_K_COUNT;
}

int lz4hip_release_workspaces(void)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(LZ4HIP_E_DEVICE, "device index out of range");
    for (HcWorkspace* pool : { g_hc_ws, g_fast_ws }) {
        HcWorkspace& w = pool[dev];
        std::unique_lock<std::mutex> lock(w.mu);
        if (pool == g_fast_ws && g_fast_slab[dev].ctl) {             // (the lane encoder's slab hangs off the same lease)
            if (w.busy && w.last) HIP_TRY(hipEventSynchronize(w.last));
            g_fast_slab[dev].release();
            w.busy = false;
        }
        if (pool == g_hc_ws && g_hc_pipe[dev].ready) {               // (the sub-chunk pipeline's streams and events hang off the LZ4HC lease)
            if (w.busy && w.last) HIP_TRY(hipEventSynchronize(w.last));
            g_hc_pipe[dev].release();
        }
        if (!w.p) continue;
        if (w.busy && w.last) HIP_TRY(hipEventSynchronize(w.last));
        HIP_TRY(hipFree(w.p));
        w.p = nullptr; w.cap = 0; w.busy = false;
    }
    {
        // the persistent lane decoder's counter ring: every slot's last user must be done (slot by slot, under the slot's lock)
        CounterRing& r = g_counter_ring[dev];
        std::lock_guard<std::mutex> lk(r.mu);
        if (r.mem) {
            for (CounterSlot& cs : r.slot) {
                std::lock_guard<std::mutex> sl(cs.mu);
                if (cs.used && cs.done) HIP_TRY(hipEventSynchronize(cs.done));
                if (cs.done) { (void)hipEventDestroy(cs.done); cs.done = nullptr; }
                cs.used = false;
            }
            HIP_TRY(hipFree(r.mem));
            r.mem = nullptr;
        }
    }
    // ... and the CALLING thread's host-pointer staging for this device (device images + pinned slots; the host-pointer
    // entry points are synchronous, so nothing of this thread's is in flight here)
    release_host_context(dev);
    // ... and that of the multi-device entry points' persistent workers for this device
    for (int k = 0; k < kMaxWorkers; k++) {
        DeviceWorker* w = nullptr;
        { std::lock_guard<std::mutex> lk(g_worker_mu); w = g_worker[k]; }
        if (!w) continue;
        std::lock_guard<std::mutex> own(w->busy);
        // (a worker is indexed by its slot, not by a device: it may hold a context for every device it ever served)
        if (w->post([] {
                int nd = 0;
                if (hipGetDeviceCount(&nd) != hipSuccess) return;
                for (int d = 0; d < nd && d < 64; d++)
                    if (host_context(d, false) && hipSetDevice(d) == hipSuccess) release_host_context(d);
            })) continue;
        w->wait();
    }
    return 0;
}

int lz4hip_tuning_set(const char* name, int value)
{
    knobs_init();
    for (int k = 0; name && k < kKnobCount; k++)
        if (strcmp(name, kKnobInfo[k].name) == 0) {
            if (value < 0 || (kKnobInfo[k].mapping && value > 2)) return fail(LZ4HIP_E_ARGUMENT, std::string("bad value for knob ") + name);
            return g_knob[k].exchange(value, std::memory_order_relaxed);
        }
    return fail(LZ4HIP_E_ARGUMENT, std::string("unknown knob ") + (name ? name : "(null)"));
}

int lz4hip_tuning_get(const char* name)
{
    knobs_init();
    // read-only: what the placement of the current device's lane-encoder slab measured, in M steps per second (0: no slab yet or not measured),
    // and how many candidate placements were built
    if (name && (strcmp(name, "encoder_slab_rate") == 0 || strcmp(name, "encoder_slab_tried") == 0 || strcmp(name, "encoder_slab_chunks") == 0)) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
        std::lock_guard<std::mutex> lk(g_fast_ws[dev].mu);
        return name[13] == 'r' ? (int)(g_fast_slab[dev].probe * 1000.0) : (name[13] == 't' ? g_fast_slab[dev].tries : (int)g_fast_slab[dev].chunks.size());
    }
    // read-only: 1 if lane-mapped decodes on the current device store their ring rows twice (the device's probe confirmed that out-of-range LDS
    // stores are dropped, and the knob decoder_wrapped_stores is 0), 0 if they run the wrapped-row instantiation.  Runs the probe if it has not run yet.
    if (name && strcmp(name, "decoder_dual_store") == 0) {
        int dev = 0;
        if (ensure_device() || hipGetDevice(&dev) != hipSuccess) return 0;
        return (lds_drop_confirmed(dev) && knob(kKnobDecoderWrappedStores) == 0) ? 1 : 0;
    }
    for (int k = 0; name && k < kKnobCount; k++)
        if (strcmp(name, kKnobInfo[k].name) == 0) return g_knob[k].load(std::memory_order_relaxed);
    return fail(LZ4HIP_E_ARGUMENT, std::string("unknown knob ") + (name ? name : "(null)"));
}

int lz4hip_encode_batch_device(const lz4hip_batch_t* b, int mode, void* stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    return launch_encode(b, mode, (hipStream_t)stream);
}

int lz4hip_decode_batch_device(const lz4hip_batch_t* b, int known_output_size, void* stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if ((rc = ensure_device())) return rc;
    return launch_decode(b, known_output_size, (hipStream_t)stream);
}

// A single-device host-pointer batch of kHostWorkersMinBlocks blocks or more runs as TWO staging pipelines on the same device (the persistent
// workers of the multi-device path, block i -> worker i mod 2): one pipeline's kernels and copies fill the gaps of the other's -- a 16 384-block
// D2 decode 23.6 -> 33.5 GB/s on the driver's box of round 5/6 (profiles/r06/bench_driver_style_call1.json, host_pointer_batch_multi_device).
// Knob host_workers: 1 = the calling thread's pipeline alone.
constexpr int64_t kHostWorkersMinBlocks = 8192;
int host_workers_for(const lz4hip_batch_t* b, int mode_is_hc)                  // (decode; fast encode only where the knob is set)
{
    if (!b || b->n_blocks < kHostWorkersMinBlocks || mode_is_hc) return 1;
    const int k = knob(kKnobHostWorkers);
    return k > 0 ? (k > 8 ? 8 : k) : 2;
}

// Slices of a host-pointer FAST encode: the wavefront-mapped encoder holds ten blocks per CU (workgroups of five) and a slice's kernel time goes by
// whole residency rounds -- 2 731 blocks (a sixth of 16 384) are 10.7 per CU = two rounds, 12.5 ms, where 2 560 blocks take 7.6 ms.  So a batch is cut
// into the fewest EQUAL slices of at most one round each (16 384 blocks: 7 x 2 341), and it runs as ONE pipeline: the kernels are the bottleneck, a
// second pipeline's slices only compete for the same ten places per CU (16 384 blocks 15.7-16.4 against 11.1-13.8 GB/s with two pipelines and 11.9-14.4
// with round 5's six equal slices, three runs each on one box: profiles/r06/host_encode_one_pipeline_equal_round_slices.txt; earlier forms of the rule:
// host_encode_slices_of_one_residency_round.txt, host_tapered_slices_ab.txt).
int64_t encode_host_slice_blocks(const lz4hip_batch_t* b, int mode)
{
    if (mode == LZ4HIP_MODE_HC) return kHcHostSliceBlocks;
    int dev = 0, cus = 0;
    if (!b || b->n_blocks < 4096 || hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    const int64_t round = (int64_t)2 * kEncodeBlocksPerGroup * cus, slices = (b->n_blocks + round - 1) / round;
    return (b->n_blocks + slices - 1) / slices;
}

int lz4hip_encode_batch_host(const lz4hip_batch_t* b, int mode)
{
    // (two pipelines only where the knob asks for them: see above)
    const int workers = mode == LZ4HIP_MODE_HC || knob(kKnobHostWorkers) <= 0 ? 1 : host_workers_for(b, 0);
    const int64_t slice = encode_host_slice_blocks(b, mode);
    int dev = 0;
    if (workers > 1 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64)
        return run_host_batch_multi(b, true, 1ull << dev, [mode](const lz4hip_batch_t* db, hipStream_t s) { return launch_encode(db, mode, s); }, slice, workers);
    return run_host_batch(b, true, [mode](const lz4hip_batch_t* db, hipStream_t s) { return launch_encode(db, mode, s); }, slice);
}

int lz4hip_decode_batch_host(const lz4hip_batch_t* b, int known_output_size)
{
    const int workers = host_workers_for(b, 0);
    int dev = 0;
    if (workers > 1 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64)
        return run_host_batch_multi(b, !known_output_size, 1ull << dev,
                                    [known_output_size](const lz4hip_batch_t* db, hipStream_t s) { return launch_decode(db, known_output_size, s); }, 0, workers);
    return run_host_batch(b, !known_output_size,
                          [known_output_size](const lz4hip_batch_t* db, hipStream_t s) { return launch_decode(db, known_output_size, s); });
}

int lz4hip_encode_batch_host_multi(const lz4hip_batch_t* b, int mode, uint64_t device_mask)
{
    return run_host_batch_multi(b, true, device_mask,
                                [mode](const lz4hip_batch_t* db, hipStream_t s) { return launch_encode(db, mode, s); },
                                mode == LZ4HIP_MODE_HC ? kHcHostSliceBlocks : 0);
}

int lz4hip_decode_batch_host_multi(const lz4hip_batch_t* b, int known_output_size, uint64_t device_mask)
{
    return run_host_batch_multi(b, !known_output_size, device_mask,
                                [known_output_size](const lz4hip_batch_t* db, hipStream_t s) { return launch_decode(db, known_output_size, s); });
}

int lz4hip_compress_limitedOutput(const char* source, char* dest, int isize, int maxOutputSize)
{
    return single(source, isize, dest, maxOutputSize, 0);
}
int lz4hip_compress(const char* source, char* dest, int isize)
{
    return single(source, isize, dest, lz4hip_compressBound(isize), 0);
}
int lz4hip_compressHC_limitedOutput(const char* source, char* dest, int isize, int maxOutputSize)
{
    return single(source, isize, dest, maxOutputSize, 1);
}
int lz4hip_compressHC(const char* source, char* dest, int isize)
{
    return single(source, isize, dest, lz4hip_compressBound(isize) + 1, 1);   /* original/lz4hc.c:762 */
}
int lz4hip_uncompress_bounded(const char* source, int isize, char* dest, int osize)
{
    return single(source, isize, dest, osize, 2);
}
int lz4hip_uncompress(const char* source, char* dest, int osize)
{
    if (!source || !dest) return fail(LZ4HIP_E_ARGUMENT, "NULL buffer");
    if (osize < 0) return fail(LZ4HIP_E_ARGUMENT, "negative length");
    return single(source, known_size_extent((const uint8_t*)source, osize), dest, osize, 2);
}
int lz4hip_uncompress_unknownOutputSize(const char* source, char* dest, int isize, int maxOutputSize)
{
    return single(source, isize, dest, maxOutputSize, 3);
}

int lz4hip_synth_device(int dist, uint64_t seed, uint64_t first_block, uint64_t block_step, int64_t n_blocks, void* out,
                        int64_t stride, int32_t len, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n_blocks <= 0 || len <= 0) return 0;
    if (!out || dist < 0 || dist > 3) return fail(LZ4HIP_E_ARGUMENT, "bad synth arguments");
    SynthArgs a = { (uint8_t*)out, stride, n_blocks, seed, first_block, block_step, len, dist };
    const unsigned grid = dist <= 1 ? 16384u : (unsigned)((n_blocks + 63) / 64);
    hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lz4hip_checksum_device(const void* data, const int64_t* off, int64_t stride, const int32_t* len,
                           int32_t len_all, uint64_t* sums, int64_t n_blocks, void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n_blocks <= 0) return 0;
    ChecksumArgs a = { (const uint8_t*)data, off, stride, len, len_all, sums, n_blocks };
    hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lz4hip_compare_device(const void* a, int64_t a_stride, const void* b, int64_t b_stride,
                          const int32_t* len, int32_t len_all, int64_t n_blocks, uint64_t* mismatches,
                          void* stream)
{
    int rc = ensure_device();
    if (rc) return rc;
    if (n_blocks <= 0) return 0;
    CompareArgs c = { (const uint8_t*)a, a_stride, (const uint8_t*)b, b_stride, len, len_all, n_blocks,
                      (unsigned long long*)mismatches };
    hipLaunchKernelGGL(compare_kernel, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
