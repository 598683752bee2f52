// Microbenchmark (not part of the product), third part of tools/microbench_random_sectors.hip: can a LIBRARY get the spread placement that
// microbench_random_sectors_spread.hip shows to be 22-32 % faster?  The 8 GiB slab is K separate hipMalloc chunks; between two chunk allocations
// a SPACER of S bytes is allocated, and all spacers are freed again before the measurement (their memory is the caller's again).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) chase(uint8_t* const* chunks, size_t regions_per_chunk, size_t region_bytes, int steps, uint32_t* sink)
{
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t* const p = (uint32_t*)(chunks[lane / regions_per_chunk] + (lane % regions_per_chunk) * region_bytes);
    const uint32_t words = (uint32_t)(region_bytes / 4);
    uint32_t s = (uint32_t)lane * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < steps; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t off = (s >> 4) % words;
        const uint32_t v = p[off];
        p[off] = v + (uint32_t)i;
        s ^= v * 0x9E3779B9u;
        acc += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 3000;
    const double pre_gib = argc > 2 ? atof(argv[2]) : 0;             // memory the "application" holds before the library allocates (GiB)
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int groups = prop.multiProcessorCount * 16;
    const size_t lanes = (size_t)groups * 64, region = 32768;
    uint32_t* sink = nullptr; (void)hipMalloc(&sink, 64);
    void* pre = nullptr;
    if (pre_gib > 0 && hipMalloc(&pre, (size_t)(pre_gib * (double)((size_t)1 << 30))) != hipSuccess) { printf("pre-allocation failed\n"); return 1; }
    size_t free_b = 0, total_b = 0; (void)hipMemGetInfo(&free_b, &total_b);
    printf("%s: %zu lanes x 32 KiB = 8 GiB in K chunks; application holds %.0f GiB, free %.1f GiB\n", prop.gcnArchName, lanes, pre_gib, free_b / 1073741824.0);
    struct { int k; double spacer_gib; } v[] = { { 1, 0 }, { 16, 0 }, { 16, 1 }, { 16, 2 }, { 16, 4 }, { 16, 7.5 }, { 16, 12 }, { 64, 1 }, { 64, 2 }, { 8, 8 }, { 8, 16 }, { 4, 32 }, { 1, 0 } };
    for (auto& t : v) {
        const size_t chunk_bytes = ((size_t)8 << 30) / t.k, spacer = (size_t)(t.spacer_gib * (double)((size_t)1 << 30));
        std::vector<uint8_t*> chunks, spacers;
        bool ok = true;
        hipEvent_t a0, a1; (void)hipEventCreate(&a0); (void)hipEventCreate(&a1);
        const double tm0 = 0;
        (void)tm0;
        for (int i = 0; i < t.k && ok; i++) {
            uint8_t *c = nullptr, *s = nullptr;
            if (hipMalloc(&c, chunk_bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            chunks.push_back(c);
            if (spacer && i + 1 < t.k) { if (hipMalloc(&s, spacer) != hipSuccess) { (void)hipGetLastError(); ok = false; break; } spacers.push_back(s); }
        }
        for (auto s : spacers) (void)hipFree(s);
        if (!ok) { printf("K %2d, spacers of %5.1f GiB: allocation failed\n", t.k, t.spacer_gib); for (auto c : chunks) (void)hipFree(c); continue; }
        for (auto c : chunks) (void)hipMemset(c, 1, chunk_bytes);
        uint8_t** d_chunks = nullptr; (void)hipMalloc(&d_chunks, sizeof(uint8_t*) * chunks.size());
        (void)hipMemcpy(d_chunks, chunks.data(), sizeof(uint8_t*) * chunks.size(), hipMemcpyHostToDevice);
        chase<<<groups, 64>>>(d_chunks, lanes / t.k, region, 200, sink);
        (void)hipEventRecord(a0);
        chase<<<groups, 64>>>(d_chunks, lanes / t.k, region, steps, sink);
        (void)hipEventRecord(a1); (void)hipEventSynchronize(a1);
        float ms = 0; (void)hipEventElapsedTime(&ms, a0, a1);
        printf("K %2d chunks of %6.0f MiB, spacers of %5.1f GiB (span %5.1f GiB): %8.2f ms  %6.2f G steps/s\n", t.k, chunk_bytes / 1048576.0, t.spacer_gib,
               8 + t.spacer_gib * (t.k - 1), ms, (double)lanes * steps / ms / 1e6);
        for (auto c : chunks) (void)hipFree(c);
        (void)hipFree(d_chunks);
    }
    return 0;
}
