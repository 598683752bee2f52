// Memory skeleton of the lane-per-block LZ4 decoder (not part of the product).
//
// One lane per 64 KiB block, 64 blocks per wavefront, exactly the global-memory accesses of the real kernel
// (lz4hip_decode_lane*.hpp) for fuzzer-style (D2) or record-style (D3) data -- and NO parsing:
//   * output: every lane produces ~8.9 (D2) / ~10.9 (D3) bytes per iteration; finished 64-byte lines leave cooperatively,
//     four lanes per line, two always-issued store instructions per iteration (records through LDS, as the kernel does);
//   * far matches: with the measured probability per iteration a 16-byte load from the lane's own output at op - offset,
//     offsets drawn from the measured distribution (profiles/r03/decoder_match_offsets.txt); only offsets beyond the ring
//     (`ring` bytes) are fetched, so the ring size is a parameter;
//   * input: ~4.4 (D2) / ~3.6 (D3) bytes per iteration, fetched as the variant says:
//       0  cooperative 32-byte pieces, two helper lanes per piece, one load instruction per iteration (generations 2/3)
//       1  every lane loads its own 32-byte piece (two load instructions per iteration, predicated)
//       2  every lane loads its own 16-byte piece (one load instruction)
//       3  every lane loads its own 64-byte sector (four load instructions)
//       4  every lane loads its own 128-byte line (eight load instructions)
//       5  NO staging at all: every lane loads the 16 bytes at its input cursor (unaligned) EVERY iteration and waits for
//          them at the top of the next one (vmcnt(3): the flush stores and the far fetch stay in flight); the sector is
//          expected to stay in the L2 because the lane touches it every iteration
//       6  as 5 with non-temporal flush stores (the streamed output should not push the input lines out of the L2)
//   * the loads of iteration i are waited for at the bottom of iteration i+1 (s_waitcnt vmcnt(N) with this iteration's
//     accesses still in flight) and their data is consumed.
// `valu` adds that many dependent-free v_perm_b32 per iteration and `lds` that many LDS dword exchanges (the real kernel:
// ~320 VALU, ~32 LDS instructions), `ldsbytes` of dynamic LDS set the residency.  What this measures is the CEILING the
// memory system sets for the mapping at a given residency / ring size / input scheme, and how it moves with them.
//
// usage: decode_skeleton [log2 blocks] "dist:input:ring:waves_per_cu:valu:lds:flush,..."       (dist 2 or 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../lz4net_amd/csrc/lz4hip_wave.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct alignas(16) Rec { uint32_t w[4]; };

struct Params {
    uint8_t* out; const uint8_t* in; int64_t n_blocks; uint64_t* sink;
    int dist, ring, valu, lds, flush;
};

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// offset of a match, given that it lies beyond 112 bytes (D2: measured shares beyond 128/192/256/512/1024/4096 bytes)
__device__ __forceinline__ int far_offset_d2(uint32_t r)
{
    const uint32_t u = r & 0xFFFFu, v = (r >> 4) & 0xFFFu;
    if (u < 19005u) return 112 + (int)(v % 80u);            // 29 %: 112 .. 191
    if (u < 27525u) return 192 + (int)(v % 64u);            // 13 %: 192 .. 255
    if (u < 44564u) return 256 + (int)(v % 256u);           // 26 %: 256 .. 511
    if (u < 57868u) return 512 + (int)(v % 512u);           // 20 %: 512 .. 1023
    return 1024 + (int)(v % 3072u);                         // 12 %: 1024 .. 4095
}

template <int INPUT>
__global__ void __launch_bounds__(64) skeleton(Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    Rec* const flush_rec = (Rec*)lds;                         // 32 records
    Rec* const load_rec = flush_rec + 32;                     // 32 records
    uint32_t* const stage = (uint32_t*)(load_rec + 32);       // 64 lanes x 4 dwords (landing zone of the cooperative pieces)
    uint32_t* const ballast = stage + 256;                    // 64 lanes x 8 dwords
    const int lane = (int)threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x * 64 + lane;
    const bool active = blk < p.n_blocks;
    uint8_t* const dst = p.out + (active ? blk : 0) * 65536;
    const uint8_t* const src = p.in + (active ? blk : 0) * 32768;
    const bool d3 = p.dist == 3;
    const int in_total = d3 ? 21600 : 32000;
    uint32_t s = (uint32_t)blk * 2654435761u + 777u, acc = 0;
    int op = 0, fl = 0, ip = 0, have = 0, pend_a = 0, pend_b = 0, done = active ? 0 : 1;
    int far_prev = 0, far_left = 0;
    wv::u32x4 fa = {0,0,0,0}, fb = {0,0,0,0}, ia[8] = {}, ib[8] = {};
    constexpr int NL = INPUT == 0 || INPUT == 2 || INPUT >= 5 ? 1 : (INPUT == 1 ? 2 : (INPUT == 3 ? 4 : 8));   // input load instructions per iteration
    constexpr int kVm = 3 + NL;
    const int piece = INPUT == 2 ? 16 : (INPUT == 3 ? 64 : (INPUT == 4 ? 128 : 32));

    auto iteration = [&](wv::u32x4& ldF, wv::u32x4& usF, wv::u32x4* ldI, wv::u32x4* usI, int& ld_pend, int& us_pend) __attribute__((always_inline)) -> bool {
        const uint32_t r = lcg(s);
        if (INPUT >= 5) {
            // the view requested in the previous iteration (first access after that iteration's top wait): three younger accesses may stay in flight
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(usI[0]) :: "memory");
            acc ^= usI[0].x ^ usI[0].w;
            wv::vm_load16_pred<0>(done == 0, (uint64_t)src + (uint64_t)(uint32_t)(ip < in_total - 16 ? ip : in_total - 16), ldI[0]);
        }
        // ---- flush: cooperative, two store instructions ----
        {
            const int unit = p.flush;                                   // 64: one line per record (four lanes); 128: two adjacent lines per record (eight lanes)
            const bool need = (done == 0) & (op - fl >= unit);
            const uint64_t needy = wv::ballot(need);
            const int cnt_all = wv::popc64(needy);
            const bool go = cnt_all >= (unit == 64 ? 16 : 8) || wv::any(need & (op - fl >= unit + 16));
            int cnt = 0; bool mine = false;
            if (go) {
                const int cap = unit == 64 ? 32 : 16;
                cnt = cnt_all < cap ? cnt_all : cap;
                const int frank = wv::rank_below(needy);
                mine = need & (frank < cap);
                if (mine) { const uint64_t dp = (uint64_t)dst; flush_rec[frank] = Rec{ { acc, (uint32_t)fl, (uint32_t)dp, (uint32_t)(dp >> 32) } }; }
                wv::mem_sync();
            }
            const int sub = unit == 64 ? (lane & 3) : (lane & 7);
#pragma unroll
            for (int base = 0; base < 32; base += 16) {
                const int idx = unit == 64 ? base + (lane >> 2) : (base >> 1) + (lane >> 3);
                const bool act = idx < cnt;
                uint64_t g = 0; uint32_t q = 0;
                if (act) { const Rec rr = flush_rec[idx]; g = ((uint64_t)rr.w[2] | ((uint64_t)rr.w[3] << 32)) + (uint64_t)(rr.w[1] + 16u * (uint32_t)sub); q = rr.w[0]; }
                if (INPUT == 6) {
                    const uint64_t m_ = __builtin_amdgcn_ballot_w64(act); uint64_t sv_; const wv::u32x4 v_ = { q, q + 1, q + 2, q + 3 };
                    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_store_dwordx4 %[a], %[d], off nt\n\ts_mov_b64 exec, %[sv]" : [sv] "=&s"(sv_) : [a] "v"(g), [d] "v"(v_), [m] "s"(m_) : "memory");
                } else
                wv::vm_store16_pred(act, g, q, q + 1, q + 2, q + 3);
            }
            if (go) { wv::mem_sync(); fl += mine ? unit : 0; }
        }
        // ---- far fetch: one load instruction ----
        {
            bool f_do = false; int f_pos = 0;
            if (!d3) {
                const bool is_far = (r & 0xFFFFu) < 25560u;                // 0.39 far loads per iteration at a 128-byte ring (3.0 G per 2^20 blocks x 7400 iterations, profiles/r03)
                const int off = far_offset_d2(lcg(s));
                f_pos = op - off;
                f_do = (done == 0) & is_far & (off > p.ring - 16) & (f_pos >= 0) & (f_pos + 16 <= fl);
            } else {
                // a match = ~2.1 chunks of 16 bytes at consecutive addresses; a new match with probability 0.33 per iteration
                const bool start = (far_left == 0) & ((r & 0xFFFFu) < 32768u);   // ~0.33 matches per iteration, ~0.67 far loads (1989 matches x 2.1 chunks per 6000 iterations)
                if (start) { const uint32_t w = op < 32768 ? (uint32_t)op : 32768u; far_prev = op - 1 - (int)(lcg(s) % (w ? w : 1u)); far_left = 1 + (int)((r >> 16) % 3u); }
                f_pos = far_prev;
                f_do = (done == 0) & (far_left > 0) & (op - f_pos > p.ring - 16) & (f_pos >= 0) & (f_pos + 16 <= fl);
                far_prev += far_left > 0 ? 16 : 0;
                far_left -= far_left > 0 ? 1 : 0;
            }
            wv::vm_load16_pred<0>(f_do, (uint64_t)dst + (uint64_t)(uint32_t)f_pos, ldF);
        }
        // ---- input ----
        if (INPUT >= 5) {
            have = in_total;
        } else if (INPUT == 0) {
            have = ((us_pend == 0) & (ip >= have)) ? (ip & ~31) : have;
            const int ahead = have - ip;
            const bool need = (done == 0) & (us_pend == 0) & (have < in_total) & (ahead <= 32);
            const uint64_t needy = wv::ballot(need);
            const int cnt = wv::popc64(needy);
            const bool go = cnt >= 24 || wv::any(need & (ahead < 26));
            bool hv = false; uint64_t g = 0;
            ld_pend = 0;
            if (go) {
                const int rank = wv::rank_below(needy);
                if (need & (rank < 32)) { const uint64_t sp = (uint64_t)src; load_rec[rank] = Rec{ { (uint32_t)lane, (uint32_t)have, (uint32_t)sp, (uint32_t)(sp >> 32) } }; ld_pend = 1; }
                wv::mem_sync();
                const int idx = lane >> 1, sub = lane & 1;
                hv = idx < (cnt < 32 ? cnt : 32);
                if (hv) { const Rec rr = load_rec[idx]; g = ((uint64_t)rr.w[2] | ((uint64_t)rr.w[3] << 32)) + (uint64_t)rr.w[1] + (uint64_t)(16 * sub); }
                wv::mem_sync();
            }
            wv::vm_load16_pred<0>(hv, g, ldI[0]);
        } else {
            const bool need = (done == 0) & (us_pend == 0) & (have < in_total) & (have - ip <= (INPUT >= 3 ? 32 : 24));
            const uint64_t g = (uint64_t)src + (uint64_t)(uint32_t)have;
            ld_pend = need ? 1 : 0;
#pragma unroll
            for (int k = 0; k < NL; k++) wv::vm_load16_pred<0>(need, g + 16 * k, ldI[k]);
        }
        // ---- ballast: what the parse / appends cost ----
        {
            uint32_t a0 = acc, a1 = acc ^ r, a2 = r, a3 = s;
            for (int k = 0; k < p.valu; k += 4) {
                a0 = wv::perm(a0, a1, 0x04020601u); a1 = wv::perm(a1, a2, 0x05030700u); a2 = wv::perm(a2, a3, 0x01040602u); a3 = wv::perm(a3, a0, 0x06000503u);
            }
            for (int k = 0; k < p.lds; k += 2) {
                ballast[((k & 7) << 6) + lane] = a0 + (uint32_t)k;
                a1 ^= ballast[(((k + 5) & 7) << 6) + lane];
            }
            acc = a0 ^ a1 ^ a2 ^ a3;
        }
        // ---- bottom: the previous iteration's loads have landed ----
        if (INPUT >= 5) asm volatile("s_waitcnt vmcnt(4)" : "+v"(usF) :: "memory");
        else wv::vm_wait<kVm>(usF, usI[0]);
#pragma unroll
        for (int k = 1; k < NL; k++) asm volatile("" : "+v"(usI[k]));
        acc ^= usF.x + usF.w;
        if (INPUT == 0) {
            if (wv::any(us_pend != 0)) { stage[lane] = usI[0].x; stage[64 + lane] = usI[0].y; stage[128 + lane] = usI[0].z; stage[192 + lane] = usI[0].w; wv::mem_sync(); acc ^= stage[(lane ^ 1) + 64]; }
        } else if (INPUT < 5) {
#pragma unroll
            for (int k = 0; k < NL; k++) acc ^= us_pend ? (usI[k].x ^ usI[k].w) : 0u;
        }
        have += us_pend ? piece : 0;
        us_pend = 0;
        // ---- progress ----
        const bool room = op - fl <= p.ring - 46;
        const bool fed = (have - ip >= 8) | (have >= in_total);
        if ((done == 0) & room & fed) {
            op += d3 ? 4 + (int)((r >> 3) % 15u) : 2 + (int)((r >> 3) % 15u);     // mean 11 / 9 bytes
            ip += d3 ? 1 + (int)((r >> 9) % 6u) : 2 + (int)((r >> 9) % 6u);       // mean 3.5 / 4.5 bytes
        }
        if ((done == 0) & (op >= 65536 - 16)) {
            // end of block: the last lines leave lane by lane
            for (; fl + 16 <= 65536; fl += 16) wv::store_global16((uint64_t)dst + (uint64_t)fl, acc, acc, acc, acc);
            done = 1;
        }
        return !wv::any(done == 0);
    };
    for (;;) {
        if (iteration(fa, fb, ia, ib, pend_a, pend_b)) break;
        if (iteration(fb, fa, ib, ia, pend_b, pend_a)) break;
    }
    if (acc == 0x12345678u) p.sink[0] = acc;
}

struct Cfg { int dist, input, ring, wpc, valu, lds, flush; };

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int64_t n = 1ll << lg;
    std::string spec = argc > 2 ? argv[2] : "2:0:128:12:0:0";
    std::vector<Cfg> cfgs;
    for (size_t pos = 0; pos < spec.size();) {
        size_t e = spec.find(',', pos); if (e == std::string::npos) e = spec.size();
        Cfg c{2, 0, 128, 12, 0, 0, 64};
        sscanf(spec.substr(pos, e - pos).c_str(), "%d:%d:%d:%d:%d:%d:%d", &c.dist, &c.input, &c.ring, &c.wpc, &c.valu, &c.lds, &c.flush);
        cfgs.push_back(c); pos = e + 1;
    }
    uint8_t *out, *in; uint64_t* sink;
    CHECK(hipMalloc(&out, (size_t)n * 65536 + 4096)); CHECK(hipMalloc(&in, (size_t)n * 32768 + 4096)); CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(out, 1, (size_t)n * 65536 + 4096)); CHECK(hipMemset(in, 2, (size_t)n * 32768 + 4096));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    printf("# decode skeleton, %lld blocks of 64 KiB; GB/s = uncompressed bytes / time; frac = (64 KiB + compressed + 8) / time / 8 TB/s\n", (long long)n);
    for (const Cfg& c : cfgs) {
        Params p{ out, in, n, sink, c.dist, c.ring, c.valu, c.lds, c.flush == 128 ? 128 : 64 };
        const size_t min_lds = 64 * 16 + 1024 + 2048;
        size_t ldsb = (size_t)(160 * 1024 / c.wpc) & ~(size_t)255;
        if (ldsb < min_lds) ldsb = min_lds;
        const unsigned grid = (unsigned)((n + 63) / 64);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a));
            switch (c.input) {
            case 0: hipLaunchKernelGGL(skeleton<0>, dim3(grid), dim3(64), ldsb, 0, p); break;
            case 1: hipLaunchKernelGGL(skeleton<1>, dim3(grid), dim3(64), ldsb, 0, p); break;
            case 2: hipLaunchKernelGGL(skeleton<2>, dim3(grid), dim3(64), ldsb, 0, p); break;
            case 3: hipLaunchKernelGGL(skeleton<3>, dim3(grid), dim3(64), ldsb, 0, p); break;
            case 4: hipLaunchKernelGGL(skeleton<4>, dim3(grid), dim3(64), ldsb, 0, p); break;
            case 5: hipLaunchKernelGGL(skeleton<5>, dim3(grid), dim3(64), ldsb, 0, p); break;
            default: hipLaunchKernelGGL(skeleton<6>, dim3(grid), dim3(64), ldsb, 0, p); break;
            }
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        const double comp = c.dist == 3 ? 21600.0 : 32000.0;
        printf("dist=%d input=%d ring=%5d flush=%3d waves/CU=%2d valu=%3d lds=%2d (LDS %6zu B): %8.3f ms  %7.1f GB/s  frac %.4f\n", c.dist, c.input, c.ring, p.flush, c.wpc, c.valu,
               c.lds, ldsb, best, (double)n * 65536 / best / 1e6, (double)n * (65536 + comp + 8) / best / 1e6 / 8000.0);
        fflush(stdout);
    }
    return 0;
}
