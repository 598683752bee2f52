// Where does a lone wavefront of the wavefront-mapped fast encoder (lz4hip_encode.hpp, encode_fast_block64k) spend its cycles?  (not part of the product)
// Each workgroup encodes one synthetic block with s_memtime accumulators around the sections of the sequence loop:
//   0 match search (wave_find_match, 64 probes per step)   1 catch-up + count + emit of the sequence a search ends in
//   2 "test next position" iterations that hit (zero-literal sequences)   3 the iteration that misses   4 (count) extra 64-probe steps
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/enc_wave_sections.hip -o tools/enc_wave_sections && tools/enc_wave_sections [blocks] [dist]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "lz4hip_wave.hpp"
__device__ unsigned long long g_cyc[8], g_cnt[8];   // (slots 4, 5 are timed in every test-next-position iteration, 6 / 2 only in those that hit, 3 in those that miss)
// (accumulated in registers, written out once per block: an atomic per section would be most of what is measured)
#define LZ4HIP_ENC_DECL() unsigned long long t_mark = 0, t_a0 = 0, t_a1 = 0, t_a2 = 0, t_a3 = 0, t_a4 = 0, t_a5 = 0, t_a6 = 0; unsigned t_c0 = 0, t_c1 = 0, t_c2 = 0, t_c3 = 0, t_c4 = 0, t_c5 = 0, t_c6 = 0
#define LZ4HIP_ENC_T0() do { t_mark = __builtin_readcyclecounter(); } while (0)
#define LZ4HIP_ENC_T(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_a##slot += now_ - t_mark; t_c##slot++; t_mark = now_; } while (0)
#define LZ4HIP_ENC_COUNT(slot, n) ((void)0)
#define LZ4HIP_ENC_FLUSH() do { if (threadIdx.x == 0) { atomicAdd(&g_cyc[0], t_a0); atomicAdd(&g_cyc[1], t_a1); atomicAdd(&g_cyc[2], t_a2); atomicAdd(&g_cyc[3], t_a3); atomicAdd(&g_cyc[4], t_a4); atomicAdd(&g_cyc[5], t_a5); atomicAdd(&g_cyc[6], t_a6); atomicAdd(&g_cnt[4], (unsigned long long)t_c4); atomicAdd(&g_cnt[5], (unsigned long long)t_c5); atomicAdd(&g_cnt[6], (unsigned long long)t_c6); \
    atomicAdd(&g_cnt[0], (unsigned long long)t_c0); atomicAdd(&g_cnt[1], (unsigned long long)t_c1); atomicAdd(&g_cnt[2], (unsigned long long)t_c2); atomicAdd(&g_cnt[3], (unsigned long long)t_c3); } } while (0)
#include "lz4hip_common.hpp"
#include "lz4hip_encode.hpp"
#include "lz4hip_synth.hpp"
using namespace lz4hip;

__global__ void __launch_bounds__(64) enc(const uint8_t* raw, uint8_t* comp, int* res, unsigned long long* total)
{
    LZ4HIP_DYN_LDS(lds);
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int r = encode_fast_block64k(raw + (size_t)blockIdx.x * 65536, 65536, comp + (size_t)blockIdx.x * 65824, 65809, lds, false);
    if (threadIdx.x == 0) { res[blockIdx.x] = r; atomicAdd(total, __builtin_readcyclecounter() - t0); }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, dist = argc > 2 ? atoi(argv[2]) : 2;
    uint8_t *raw, *comp; int* res; unsigned long long* total;
    hipMalloc(&raw, (size_t)n * 65536); hipMalloc(&comp, (size_t)n * 65824); hipMalloc(&res, n * 4); hipMalloc(&total, 8);
    SynthArgs a = { raw, 65536, n, 20260925ull, 0, 1, 65536, dist };
    hipLaunchKernelGGL(synth_kernel, dim3(dist <= 1 ? 4096 : (n + 63) / 64), dim3(64), 0, 0, a);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long z[8] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof z); hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), z, sizeof z); hipMemset(total, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(enc, dim3(n), dim3(64), kFastTableBytes, 0, raw, comp, res, total);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc[8], cnt[8], tot;
        hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof cyc); hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof cnt); hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost);
        std::vector<int> r(n); hipMemcpy(r.data(), res, n * 4, hipMemcpyDeviceToHost);
        long long csum = 0; for (int v : r) csum += v;
        if (rep == 0) continue;
        printf("dist %d, %d blocks (one wavefront each), kernel %.3f ms, mean compressed %.0f B; cycle counter ticks per block %.0f\n", dist, n, ms, (double)csum / n, (double)tot / n);
        const char* name[7] = { "match search (wave_find_match)", "catch-up + count + emit after a search", "test-next-position that hits: emit (after the count)", "test-next-position that misses: after the table",
                                "test-next-position: checks, loads issued, window, hashes", "test-next-position: table put / get / put", "test-next-position that hits: candidate load, test + count" };
        for (int k = 0; k < 7; k++)
            printf("  %-44s %9.0f per block x %7.0f ticks = %5.1f %% of the block's ticks\n", name[k], (double)cnt[k] / n, cnt[k] ? (double)cyc[k] / cnt[k] : 0.0, 100.0 * cyc[k] / (double)tot);
    }
    return 0;
}
