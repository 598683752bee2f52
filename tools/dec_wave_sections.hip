// Where does a lone wavefront of the wavefront-mapped decoder (lz4hip_decode.hpp, decode_block<true>) spend its cycles?  (not part of the product)
// Sections of its sequence loop, s_memtime ticks accumulated in registers:
//   0 a burst that came about (several short sequences, <= 64 output bytes)   1 a burst attempt that did not   2 the common short sequence in one step
//   3 the general path (one sequence)      counts: 4 sequences per burst, 5 dependency rounds per burst
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/dec_wave_sections.hip -o tools/dec_wave_sections && tools/dec_wave_sections [blocks] [dist]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "lz4hip_wave.hpp"
__device__ unsigned long long g_cyc[8], g_cnt[8];
#define LZ4HIP_DEC_DECL() unsigned long long t_mark = 0, t_a0 = 0, t_a1 = 0, t_a2 = 0, t_a3 = 0; unsigned t_c0 = 0, t_c1 = 0, t_c2 = 0, t_c3 = 0, t_c4 = 0, t_c5 = 0
#define LZ4HIP_DEC_T0() do { t_mark = __builtin_readcyclecounter(); } while (0)
#define LZ4HIP_DEC_T(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_a##slot += now_ - t_mark; t_c##slot++; t_mark = now_; } while (0)
#define LZ4HIP_DEC_ADD(slot, n) do { t_c##slot += (unsigned)(n); } while (0)
#define LZ4HIP_DEC_FLUSH() do { if (threadIdx.x % 64 == 0) { atomicAdd(&g_cyc[0], t_a0); atomicAdd(&g_cyc[1], t_a1); atomicAdd(&g_cyc[2], t_a2); atomicAdd(&g_cyc[3], t_a3); \
    atomicAdd(&g_cnt[0], (unsigned long long)t_c0); atomicAdd(&g_cnt[1], (unsigned long long)t_c1); atomicAdd(&g_cnt[2], (unsigned long long)t_c2); atomicAdd(&g_cnt[3], (unsigned long long)t_c3); \
    atomicAdd(&g_cnt[4], (unsigned long long)t_c4); atomicAdd(&g_cnt[5], (unsigned long long)t_c5); } } while (0)
#include "lz4hip_common.hpp"
#include "lz4hip_decode.hpp"
#include "lz4hip_encode.hpp"
#include "lz4hip_synth.hpp"
using namespace lz4hip;

__global__ void __launch_bounds__(64) enc(const uint8_t* raw, uint8_t* comp, int* res)
{
    LZ4HIP_DYN_LDS(lds);
    const int r = encode_fast_block64k(raw + (size_t)blockIdx.x * 65536, 65536, comp + (size_t)blockIdx.x * 65824, 65809, lds, false);
    if (threadIdx.x == 0) res[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) dec(const uint8_t* comp, const int* clen, uint8_t* back, int* res, unsigned long long* total, int n)
{
    LZ4HIP_STATIC_LDS(rings, 4 * kWaveLdsBytes);
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= n) return;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int r = decode_block<true>(comp + (size_t)blk * 65824, wv::uniform(clen[blk]), back + (size_t)blk * 65536, 65536, rings + (threadIdx.x >> 6) * kWaveLdsBytes);
    if (threadIdx.x % 64 == 0) { res[blk] = r; atomicAdd(total, __builtin_readcyclecounter() - t0); }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1024, dist = argc > 2 ? atoi(argv[2]) : 2;
    uint8_t *raw, *comp, *back; int *clen, *res; unsigned long long* total;
    hipMalloc(&raw, (size_t)n * 65536); hipMalloc(&back, (size_t)n * 65536); hipMalloc(&comp, (size_t)n * 65824); hipMalloc(&clen, n * 4); hipMalloc(&res, n * 4); hipMalloc(&total, 8);
    SynthArgs a = { raw, 65536, n, 20260925ull, 0, 1, 65536, dist };
    hipLaunchKernelGGL(synth_kernel, dim3(dist <= 1 ? 4096 : (n + 63) / 64), dim3(64), 0, 0, a);
    hipLaunchKernelGGL(enc, dim3(n), dim3(64), kFastTableBytes, 0, raw, comp, clen);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long z[8] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof z); hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), z, sizeof z); hipMemset(total, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(dec, dim3((n + 3) / 4), dim3(256), 0, 0, comp, clen, back, res, total, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc[8], cnt[8], tot;
        hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof cyc); hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof cnt); hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost);
        std::vector<int> r(n), c(n); hipMemcpy(r.data(), res, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), clen, n * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < n; i++) bad += r[i] != c[i];
        if (rep == 0) continue;
        printf("dist %d, %d blocks (one wavefront each, four per workgroup), kernel %.3f ms, results ok %s; cycle counter ticks per block %.0f\n", dist, n, ms, bad ? "NO" : "yes", (double)tot / n);
        const char* name[4] = { "burst that came about", "burst attempt that did not", "short sequence in one step", "general path (one sequence)" };
        for (int k = 0; k < 4; k++)
            printf("  %-32s %8.0f per block x %7.0f ticks = %5.1f %% of the block's ticks\n", name[k], (double)cnt[k] / n, cnt[k] ? (double)cyc[k] / cnt[k] : 0.0, 100.0 * cyc[k] / (double)tot);
        printf("  sequences per burst %.2f, dependency rounds per burst %.2f\n", cnt[0] ? (double)cnt[4] / cnt[0] : 0.0, cnt[0] ? (double)cnt[5] / cnt[0] : 0.0);
    }
    return 0;
}
