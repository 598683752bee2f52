// simt_wave.hpp -- TEST INFRASTRUCTURE ONLY: CPU emulation of lz4net_amd/csrc/lz4hip_wave.hpp
// (same `wv::` API, implemented on the fiber scheduler of simt.hpp).  Besides emulating the
// semantics it CHECKS the uniformity claims the kernels make (wv::uniform / readlane index).
#pragma once
#include "simt.hpp"

#define LZ4HIP_WAVE_API 1
#define LZ4HIP_DEVICE inline
#define LZ4HIP_DYN_LDS(name) unsigned char* name = simt::rt().lds.data()
#define LZ4HIP_STATIC_LDS(name, bytes) unsigned char* name = simt::rt().lds.data()
#define LZ4HIP_KEEP(x) ((void)0)

namespace wv {

constexpr int kWave = 64;

inline int lane() { return simt::rt().cur->tid & 63; }
inline int wave_in_block() { return simt::rt().cur->tid >> 6; }

inline uint64_t uniform64(uint64_t v, int site)
{
    unsigned p, tag;
    const simt::Lane* w = simt::exchange(v, site, &p, &tag);
    const int n = simt::wave_width(simt::rt().cur);
    for (int i = 0; i < n; i++)
        if (simt::took_part(w[i], p, tag)) {
            if (w[i].slot[p] != v) simt::die("wv::uniform() on a value that is not wave-uniform", site, i);
            return w[i].slot[p];
        }
    return v;
}
inline uint32_t uniform(uint32_t v) { return (uint32_t)uniform64(v, 101); }
inline int32_t uniform(int32_t v) { return (int32_t)uniform64((uint32_t)v, 102); }
inline uint64_t uniform(uint64_t v) { return uniform64(v, 103); }
inline int64_t uniform(int64_t v) { return (int64_t)uniform64((uint64_t)v, 104); }

inline uint64_t first_lane(uint64_t v)
{
    unsigned p, tag;
    const simt::Lane* w = simt::exchange(v, 105, &p, &tag);
    const int n = simt::wave_width(simt::rt().cur);
    for (int i = 0; i < n; i++)
        if (simt::took_part(w[i], p, tag)) return w[i].slot[p];
    return v;
}

inline uint32_t readlane(uint32_t v, int src_lane)
{
    unsigned p, tag;
    // the index must be wave-uniform on hardware (it is an SGPR operand): check it
    const simt::Lane* w = simt::exchange(((uint64_t)(uint32_t)src_lane << 32) | v, 110, &p, &tag);
    const int n = simt::wave_width(simt::rt().cur);
    for (int i = 0; i < n; i++)
        if (simt::took_part(w[i], p, tag) && (int)(w[i].slot[p] >> 32) != src_lane)
            simt::die("wv::readlane() with a non-uniform lane index", src_lane, (int)(w[i].slot[p] >> 32));
    if (src_lane < 0 || src_lane >= 64) simt::die("wv::readlane() lane out of range", src_lane);
    return (uint32_t)w[src_lane].slot[p];
}

inline uint32_t scan_add(uint32_t x)
{
    unsigned p, tag;
    const simt::Lane* w = simt::exchange(x, 125, &p, &tag);
    const int n = simt::wave_width(simt::rt().cur), me = lane();
    uint32_t sum = 0;
    for (int i = 0; i < n && i <= me; i++) {
        if (!simt::took_part(w[i], p, tag)) simt::die("wv::scan_add() with inactive lanes (the DPP scan needs all 64)", i);
        sum += (uint32_t)w[i].slot[p];
    }
    return sum;
}

inline uint32_t writelane(uint32_t old, uint32_t value, int dst_lane)
{
    // value and lane index are SGPR operands on hardware: both must be wave-uniform
    if (uniform64(((uint64_t)(uint32_t)dst_lane << 32) | value, 115) != ((((uint64_t)(uint32_t)dst_lane) << 32) | value)) simt::die("wv::writelane() with non-uniform operands", dst_lane);
    if (dst_lane < 0 || dst_lane >= 64) simt::die("wv::writelane() lane out of range", dst_lane);
    return lane() == dst_lane ? value : old;
}

inline uint32_t shuffle(uint32_t v, int src_lane)
{
    unsigned p, tag;
    const simt::Lane* w = simt::exchange(v, 120, &p, &tag);
    return (uint32_t)w[src_lane & 63].slot[p];
}

inline uint64_t ballot(bool pred)
{
    unsigned p, tag;
    const simt::Lane* w = simt::exchange(pred ? 1 : 0, 130, &p, &tag);
    const int n = simt::wave_width(simt::rt().cur);
    uint64_t m = 0;
    for (int i = 0; i < n; i++)
        if (simt::took_part(w[i], p, tag) && w[i].slot[p]) m |= 1ull << i;
    return m;
}
inline bool any(bool pred) { return ballot(pred) != 0; }
inline int rank_below(uint64_t m) { return __builtin_popcountll(m & ((1ull << lane()) - 1ull)); }

inline void mem_sync() { simt::yield(simt::WAIT_WAVE, 140); }
inline void block_sync() { simt::yield(simt::WAIT_BLOCK, 150); }
inline void lds_barrier() { simt::yield(simt::WAIT_BLOCK, 150); }
// s_waitcnt vmcnt(0): every access issued through vm_load16_* / vm_store16_* so far has finished
inline void wait_vector_memory()
{
    auto& q = simt::rt().cur->vmq;
    for (const simt::PendingVm& op : q)
        if (op.is_load && op.addr) __builtin_memcpy(op.dst, (const void*)op.addr, 16);
    q.clear();
}

inline uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int b = 0; b < 4; b++) {
        const uint32_t s = (sel >> (8 * b)) & 255u;
        uint32_t byte;
        if (s < 8) byte = (uint32_t)(v >> (8 * s)) & 255u;
        else if (s == 0x0C) byte = 0;
        else { simt::die("wv::perm() selector outside 0..7 / 0x0C", (int)s); byte = 0; }
        r |= byte << (8 * b);
    }
    return r;
}
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
    if (n > 3) simt::die("wv::alignbyte() shift outside 0..3 (hardware behaviour differs between ISA revisions)", (int)n);
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * n));
}

inline void store_global16(uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t v[4] = { a, b, c, d };
    __builtin_memcpy((void*)addr, v, 16);
}

inline void load_global16(uint64_t addr, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    uint32_t v[4];
    __builtin_memcpy(v, (const void*)addr, 16);
    a = v[0]; b = v[1]; c = v[2]; d = v[3];
}

inline void store16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t v[4] = { a, b, c, d };
    __builtin_memcpy(p, v, 16);
}
inline void load16(const void* p, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d)
{
    uint32_t v[4];
    __builtin_memcpy(v, p, 16);
    a = v[0]; b = v[1]; c = v[2]; d = v[3];
}

// ---- hand-counted vector memory (lz4hip_decode_lane3.hpp) ----
// The emulation is ASYNCHRONOUS on purpose: a load only writes its destination when a vm_wait<N> retires it (and reads
// memory at that moment, the latest the hardware could), so a kernel that reads the destination registers before the
// wait that covers them, or miscounts N, sees stale data here too.  Every call counts as one instruction for every lane
// of the wavefront (the hardware issues it with an empty mask).
struct u32x4 { uint32_t x, y, z, w; };
template <int POLICY = 0>
inline void vm_load16_pred(bool pred, uint64_t addr, u32x4& v)
{
    simt::rt().cur->vmq.push_back(simt::PendingVm{ pred ? addr : 0, (uint32_t*)&v, true });
}
inline void vm_store16_pred(bool pred, uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    if (pred) { const uint32_t v[4] = { a, b, c, d }; __builtin_memcpy((void*)addr, v, 16); }
    simt::rt().cur->vmq.push_back(simt::PendingVm{ 0, nullptr, false });
}
template <int N>
inline void vm_wait(u32x4&, u32x4&)
{
    auto& q = simt::rt().cur->vmq;
    while ((int)q.size() > N) {
        const simt::PendingVm op = q.front();
        q.erase(q.begin());
        if (op.is_load && op.addr) __builtin_memcpy(op.dst, (const void*)op.addr, 16);
    }
}
template <int N, int M>
inline void vm_wait_list(u32x4& a, u32x4 (&l)[M]) { u32x4 dummy{}; vm_wait<N>(a, dummy); }
struct mask_t { bool v; };
inline mask_t cond(bool p) { return mask_t{ p }; }
inline mask_t operator&(mask_t a, mask_t b) { return mask_t{ a.v && b.v }; }
inline mask_t operator|(mask_t a, mask_t b) { return mask_t{ a.v || b.v }; }
inline mask_t operator~(mask_t a) { return mask_t{ !a.v }; }
inline bool any(mask_t m) { return any(m.v); }
inline uint32_t sel(mask_t m, uint32_t a, uint32_t b) { return m.v ? a : b; }
template <int POLICY = 0>
inline void vm_load16_mask(mask_t m, uint64_t addr, u32x4& v) { vm_load16_pred<POLICY>(m.v, addr, v); }
template <int POLICY, int OFF>
inline void vm_load16_mask_off(mask_t m, uint64_t addr, u32x4& v) { vm_load16_pred<POLICY>(m.v, addr + (uint64_t)OFF, v); }
inline void vm_store16_mask(mask_t m, uint64_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { vm_store16_pred(m.v, addr, a, b, c, d); }
// (the hardware drops a store outside the workgroup's allocation and returns 0 for such a load; the address arithmetic is 32-bit)
template <int OFF>
inline void lds_store_drop(unsigned char* lds, uint32_t lds_bytes, uint32_t a, uint32_t v)
{
    const uint32_t t = a + (uint32_t)OFF;
    if (t <= lds_bytes - 4u) std::memcpy(lds + t, &v, 4);
}
template <int ROW0, int ROW1>
inline void lds_store2_rows_drop(unsigned char* lds, uint32_t lds_bytes, uint32_t a, uint32_t v0, uint32_t v1)
{
    lds_store_drop<256 * ROW0>(lds, lds_bytes, a, v0);
    lds_store_drop<256 * ROW1>(lds, lds_bytes, a, v1);
}
template <int OFF>
inline uint32_t lds_load_zero(const unsigned char* lds, uint32_t lds_bytes, uint32_t a)
{
    const uint32_t t = a + (uint32_t)OFF;
    uint32_t r = 0;
    if (t <= lds_bytes - 4u) std::memcpy(&r, lds + t, 4);
    return r;
}
inline void lds_mskor(uint32_t* p, uint32_t mask, uint32_t data) { *p = (*p & ~mask) | data; }

inline int ctz64(uint64_t m) { return __builtin_ctzll(m); }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }

}  // namespace wv
