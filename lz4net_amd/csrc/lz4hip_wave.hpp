// lz4hip_wave.hpp -- gfx950 wavefront primitives used by every lz4hip kernel.
//
// A 64-lane wavefront is the unit that owns one LZ4 block.  Control state of the (inherently
// sequential) LZ4 parse is kept wave-uniform in SGPRs; the 64 lanes are used for data movement,
// match counting (ballot + ctz) and candidate evaluation.  This header is the ONLY place that names
// AMDGCN builtins; kernels are written against the `wv::` API below.  (tests/simt/ provides a
// CPU emulation of the same API so the kernel sources can be exercised without a GPU; that is
// test infrastructure and is never linked into liblz4hip.so.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LZ4HIP_WAVE_API 1
#define LZ4HIP_DEVICE __device__ __forceinline__
// Dynamic LDS of the current workgroup, 16-byte aligned (cdna guide, Guideline 17).
#define LZ4HIP_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// Statically sized LDS of the current workgroup: its address is a compile-time constant that folds into the
// offset field of ds_* instructions (a dynamic array costs one v_add per computed address).
#define LZ4HIP_STATIC_LDS(name, bytes) __shared__ __attribute__((aligned(16))) unsigned char name[bytes]
// Forces `x` to be computed here (stops the compiler from sinking a load into a later conditional block).
#define LZ4HIP_KEEP(x) asm volatile("" : "+v"(x))

namespace wv {

constexpr int kWave = 64;

LZ4HIP_DEVICE int lane() { return (int)(threadIdx.x & 63u); }
LZ4HIP_DEVICE int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Promote a value the programmer knows to be wave-uniform into an SGPR.
LZ4HIP_DEVICE uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
LZ4HIP_DEVICE int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
LZ4HIP_DEVICE uint64_t uniform(uint64_t v)
{
    uint32_t lo = uniform((uint32_t)v), hi = uniform((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
LZ4HIP_DEVICE int64_t uniform(int64_t v) { return (int64_t)uniform((uint64_t)v); }

// Broadcast of the first active lane's value (v_readfirstlane); unlike uniform() the lanes may disagree.
LZ4HIP_DEVICE uint64_t first_lane(uint64_t v) { return uniform(v); }

// v_readlane_b32 with a wave-uniform lane index: result lands in an SGPR.
LZ4HIP_DEVICE uint32_t readlane(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane); }

// v_writelane_b32 with a wave-uniform value and a wave-uniform lane index: lane `dst_lane` of the result holds `value`, every
// other lane keeps `old`.
LZ4HIP_DEVICE uint32_t writelane(uint32_t old, uint32_t value, int dst_lane)
{
    // (inline assembly: this compiler has no writelane builtin.  gfx9 allows ONE scalar register per vector instruction, so the
    //  lane select travels in M0; the s_nop covers the wait states a lane select needs after its register was written, which the
    //  compiler cannot see into; readfirstlane pins both operands to scalar registers)
    const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)value);
    const int sl = __builtin_amdgcn_readfirstlane(dst_lane);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(sv), "s"(sl) : "m0");
    return old;
}

// Inclusive prefix sum over the 64 lanes (all lanes active): five DPP row shifts / broadcasts, no LDS, no scalar round trip.
LZ4HIP_DEVICE uint32_t scan_add(uint32_t x)
{
#define LZ4HIP_DPP_ADD(CTRL, ROWS) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xF, false)
    LZ4HIP_DPP_ADD(0x111, 0xF);      // row_shr:1
    LZ4HIP_DPP_ADD(0x112, 0xF);      // row_shr:2
    LZ4HIP_DPP_ADD(0x114, 0xF);      // row_shr:4
    LZ4HIP_DPP_ADD(0x118, 0xF);      // row_shr:8
    LZ4HIP_DPP_ADD(0x142, 0xA);      // row_bcast:15 into rows 1 and 3
    LZ4HIP_DPP_ADD(0x143, 0xC);      // row_bcast:31 into rows 2 and 3
#undef LZ4HIP_DPP_ADD
    return x;
}

// Arbitrary cross-lane gather (ds_bpermute_b32): lane i receives v from lane idx_i.
LZ4HIP_DEVICE uint32_t shuffle(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }

LZ4HIP_DEVICE uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // (takes the lane mask the compare produced; __ballot() goes through an integer)
LZ4HIP_DEVICE bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// number of set bits of a wave-uniform mask below this lane's bit (v_mbcnt_lo/hi)
LZ4HIP_DEVICE int rank_below(uint64_t m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// Orders this wave's earlier memory operations before its later ones as seen by the OTHER lanes of
// the same wave.  The hardware already executes a wave's vector-memory instructions in order; this
// only stops the compiler from moving a load above a store it cannot see a same-thread dependence on.
LZ4HIP_DEVICE void mem_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

LZ4HIP_DEVICE void block_sync() { __syncthreads(); }
// Workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every global access the wavefront has in
// flight (vmcnt(0)), which a kernel that keeps loads and stores in flight across its barriers must not pay.
LZ4HIP_DEVICE void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(0): every vector-memory access this wave has issued has finished.  Used right after a RARE load whose
// destination registers are read in a hot loop: the compiler otherwise has to assume at the loop header that the load may
// still be in flight and puts the wait (which on gfx9 also covers every store issued since) in front of the hot reads.
LZ4HIP_DEVICE void wait_vector_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// v_perm_b32: result byte i = byte sel.byte[i] of the 8-byte value {hi, lo} (0..3 -> lo, 4..7 -> hi; 0x0C -> 0x00).
LZ4HIP_DEVICE uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_alignbyte_b32: ({hi, lo} >> 8 * n) truncated to 32 bits; callers keep n in 0..3.
LZ4HIP_DEVICE uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }

// 16-byte store to a GLOBAL address rebuilt from integers, any alignment (global_store_dwordx4, not flat_*).
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
LZ4HIP_DEVICE void store_global16(uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    u32x4_unaligned v = { a, b, c, d };
    *(__attribute__((address_space(1))) u32x4_unaligned*)addr = v;
}

// 16-byte load from a GLOBAL address rebuilt from integers (global_load_dwordx4: a flat_load would also count on
// lgkmcnt and make the next LDS access wait for it).
LZ4HIP_DEVICE void load_global16(uint64_t addr, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    const u32x4_unaligned v = *(const __attribute__((address_space(1))) u32x4_unaligned*)addr;
    a = v.x; b = v.y; c = v.z; d = v.w;
}

// 16 bytes to / from any address, any alignment, as ONE dwordx4 access (a plain 4-byte-aligned struct copy is split
// into four dword accesses by the compiler).
LZ4HIP_DEVICE void store16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    u32x4_unaligned v = { a, b, c, d };
    *(u32x4_unaligned*)p = v;
}
LZ4HIP_DEVICE void load16(const void* p, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    const u32x4_unaligned v = *(const u32x4_unaligned*)p;
    a = v.x; b = v.y; c = v.z; d = v.w;
}

// ---- hand-counted vector memory (lz4hip_decode_lane3.hpp) ---------------------------------------------------------------
// A loop that issues the SAME NUMBER of vector-memory instructions every iteration can wait with s_waitcnt vmcnt(N) for
// the loads of the PREVIOUS iteration while this iteration's are still in flight (gfx9 returns loads and stores of a
// wavefront in issue order and counts both in vmcnt).  The compiler only does that counting in straight-line code, so
// these accesses are inline assembly: predicated by an exec mask INSIDE the asm (the instruction is always issued, with an
// empty mask if need be -- it still counts), invisible to the compiler's own s_waitcnt insertion, and ordered against
// their users by vm_wait<N>(), which takes the destination registers as in/out operands.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// POLICY: 0 = default cache policy, 1 = nt (non-temporal: the line is the first to go from the L2), 2 = sc1, 3 = sc0 sc1
// (system scope: served by the memory side without a place in the L2) -- the last two only in tuning builds' sweeps
template <int POLICY = 0>
LZ4HIP_DEVICE void vm_load16_pred(bool pred, uint64_t addr, u32x4& v)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
    uint64_t saved;
#define LZ4HIP_VM_LOAD(MODS)                                                                                                      \
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dwordx4 %[d], %[a], off" MODS "\n\ts_mov_b64 exec, %[sv]"       \
                 : [d] "+v"(v), [sv] "=&s"(saved) : [a] "v"(addr), [m] "s"(m) : "memory")
    if (POLICY == 1) LZ4HIP_VM_LOAD(" nt");
    else if (POLICY == 2) LZ4HIP_VM_LOAD(" sc1");
    else if (POLICY == 3) LZ4HIP_VM_LOAD(" sc0 sc1");
    else LZ4HIP_VM_LOAD("");
#undef LZ4HIP_VM_LOAD
}
LZ4HIP_DEVICE void vm_store16_pred(bool pred, uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(pred);
    uint64_t saved;
    const u32x4 v = { a, b, c, d };
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_store_dwordx4 %[a], %[d], off\n\ts_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved) : [a] "v"(addr), [d] "v"(v), [m] "s"(m) : "memory");
}
// At most N of this wavefront's vector-memory instructions are still in flight afterwards; `a` and `b` (destinations of
// vm_load16_pred) may not be read before.
template <int N>
LZ4HIP_DEVICE void vm_wait(u32x4& a, u32x4& b)
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(a), "+v"(b) : [n] "n"(N) : "memory");
}

// The same wait for a loop that keeps more load destinations: `a` and every element of `l` may not be read before.
template <int N, int M>
LZ4HIP_DEVICE void vm_wait_list(u32x4& a, u32x4 (&l)[M])
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(a) : [n] "n"(N) : "memory");
#pragma unroll
    for (int j = 0; j < M; j++) asm volatile("" : "+v"(l[j]));
}

// ---- per-lane selects that STAY selects (lz4hip_decode_lane4.hpp) ----------------------------------------------------------
// A binary tree of `c ? a[i + k] : a[i]` over a register array is how a lane takes bytes at a per-lane position out of a
// window it keeps in registers.  Written in plain C++ the compiler recognises a dynamically indexed array and moves the array
// to scratch memory (measured: 4 000 cycles per lookup, tools/microbench_select_tree.hip); v_cndmask_b32 through inline
// assembly keeps it in registers (4.5 cycles per select).  cond() turns the per-lane condition into the lane mask once.
// A lane mask is a value of its own (an SGPR pair): masks of SIMPLE comparisons combine with scalar instructions.  (Handing
// a compound condition to a ballot makes the compiler materialise the bool in a VGPR and compare it again: two vector
// instructions per ballot.)
struct mask_t { uint64_t v; };
LZ4HIP_DEVICE mask_t cond(bool p) { return mask_t{ __builtin_amdgcn_ballot_w64(p) }; }
LZ4HIP_DEVICE mask_t operator&(mask_t a, mask_t b) { return mask_t{ a.v & b.v }; }
LZ4HIP_DEVICE mask_t operator|(mask_t a, mask_t b) { return mask_t{ a.v | b.v }; }
LZ4HIP_DEVICE mask_t operator~(mask_t a) { return mask_t{ ~a.v }; }
LZ4HIP_DEVICE bool any(mask_t m) { return m.v != 0ull; }
LZ4HIP_DEVICE uint32_t sel(mask_t m, uint32_t a, uint32_t b)      // m ? a : b
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m.v));
    return r;
}
// vm_load16_pred / vm_store16_pred with the lane mask given
template <int POLICY = 0>
LZ4HIP_DEVICE void vm_load16_mask(mask_t mk, uint64_t addr, u32x4& v)
{
    const uint64_t m = mk.v;
    uint64_t saved;
#define LZ4HIP_VM_LOAD(MODS)                                                                                                      \
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dwordx4 %[d], %[a], off" MODS "\n\ts_mov_b64 exec, %[sv]"       \
                 : [d] "+v"(v), [sv] "=&s"(saved) : [a] "v"(addr), [m] "s"(m) : "memory")
    if (POLICY == 1) LZ4HIP_VM_LOAD(" nt");
    else if (POLICY == 2) LZ4HIP_VM_LOAD(" sc1");
    else if (POLICY == 3) LZ4HIP_VM_LOAD(" sc0 sc1");
    else LZ4HIP_VM_LOAD("");
#undef LZ4HIP_VM_LOAD
}
// ... with a constant byte offset in the instruction's offset field (0 .. 4095): the loads of one piece share one address pair
template <int POLICY, int OFF>
LZ4HIP_DEVICE void vm_load16_mask_off(mask_t mk, uint64_t addr, u32x4& v)
{
    static_assert(OFF >= 0 && OFF < 4096, "global_load offset field");
    const uint64_t m = mk.v;
    uint64_t saved;
#define LZ4HIP_VM_LOAD(MODS)                                                                                                                      \
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dwordx4 %[d], %[a], off offset:%[o]" MODS "\n\ts_mov_b64 exec, %[sv]"       \
                 : [d] "+v"(v), [sv] "=&s"(saved) : [a] "v"(addr), [m] "s"(m), [o] "n"(OFF) : "memory")
    if (POLICY == 1) LZ4HIP_VM_LOAD(" nt");
    else if (POLICY == 2) LZ4HIP_VM_LOAD(" sc1");
    else if (POLICY == 3) LZ4HIP_VM_LOAD(" sc0 sc1");
    else LZ4HIP_VM_LOAD("");
#undef LZ4HIP_VM_LOAD
}
LZ4HIP_DEVICE void vm_store16_mask(mask_t mk, uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint64_t m = mk.v;
    uint64_t saved;
    const u32x4 v = { a, b, c, d };
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_store_dwordx4 %[a], %[d], off\n\ts_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved) : [a] "v"(addr), [d] "v"(v), [m] "s"(m) : "memory");
}

// DS_MSKOR_B32: MEM = (MEM & ~mask) | data -- a byte-granular merge into an aligned LDS dword without reading it back.
// (An LDS instruction the compiler does not know about only makes its own lgkmcnt waits stricter: LDS operations of a
// wavefront complete in issue order.)
LZ4HIP_DEVICE void lds_mskor(uint32_t* p, uint32_t mask, uint32_t data)
{
    asm volatile("ds_mskor_b32 %[a], %[m], %[d]" : : [a] "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p), [m] "v"(mask), [d] "v"(data) : "memory");
}

// DS_WRITE_B32 at lds + a + OFF where that address MAY LIE OUTSIDE the workgroup's LDS allocation of lds_bytes (a is a 32-bit value
// that may have wrapped below zero): gfx950 drops such a store -- no fault, nothing outside the allocation changes
// (tools/lds_out_of_range.hip, profiles/r04/lds_out_of_range.txt).  The lane decoder issues every ring store twice, at `row` and at
// `row - ring size`, instead of wrapping the row address with three vector-ALU instructions.
template <int OFF>
LZ4HIP_DEVICE void lds_store_drop(unsigned char* lds, uint32_t lds_bytes, uint32_t a, uint32_t v)
{
    (void)lds_bytes;
    asm volatile("ds_write_b32 %[a], %[v] offset:%[o]" : : [a] "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + a), [v] "v"(v), [o] "n"(OFF) : "memory");
}
// DS_WRITE2ST64_B32: two dwords in one instruction, at lds + a + 256 * ROW0 and lds + a + 256 * ROW1; each of the two addresses is checked
// (and dropped) on its own.
template <int ROW0, int ROW1>
LZ4HIP_DEVICE void lds_store2_rows_drop(unsigned char* lds, uint32_t lds_bytes, uint32_t a, uint32_t v0, uint32_t v1)
{
    (void)lds_bytes;
    asm volatile("ds_write2st64_b32 %[a], %[v0], %[v1] offset0:%[o0] offset1:%[o1]"
                 : : [a] "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + a), [v0] "v"(v0), [v1] "v"(v1), [o0] "n"(ROW0), [o1] "n"(ROW1) : "memory");
}
// DS_READ_B32 under the same rule: an out-of-range load returns 0 (same evidence)
template <int OFF>
LZ4HIP_DEVICE uint32_t lds_load_zero(const unsigned char* lds, uint32_t lds_bytes, uint32_t a)
{
    (void)lds_bytes;
    uint32_t r;
    asm volatile("ds_read_b32 %[r], %[a] offset:%[o]" : [r] "=v"(r) : [a] "v"((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)lds + a), [o] "n"(OFF) : "memory");
    return r;
}

LZ4HIP_DEVICE int ctz64(uint64_t m) { return __builtin_ctzll(m); }
LZ4HIP_DEVICE int popc64(uint64_t m) { return __builtin_popcountll(m); }

}  // namespace wv
