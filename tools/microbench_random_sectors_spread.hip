// Microbenchmark (not part of the product), second part of tools/microbench_random_sectors.hip: the same dependent random read-modify-write
// chains (262 144 lanes x 32 KiB regions = 8 GiB), but the 8 GiB are CHUNKS of one 128 GiB allocation taken every stride-th chunk:
// does the rate depend on how far the slab is spread over device memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void __launch_bounds__(64) chase(uint8_t* base, size_t region_bytes, size_t chunk_bytes, size_t stride, int steps, uint32_t* sink)
{
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t logical = lane * region_bytes, c = logical / chunk_bytes;
    uint32_t* const p = (uint32_t*)(base + c * chunk_bytes * stride + logical % chunk_bytes);
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
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int groups = prop.multiProcessorCount * 16;
    const size_t lanes = (size_t)groups * 64, region = 32768, total = (size_t)128 << 30;
    uint8_t* slab = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&slab, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(slab, 1, total);
    (void)hipDeviceSynchronize();
    printf("%s: %zu lanes x 32 KiB = 8 GiB of a 128 GiB allocation at %p, %d steps per lane\n", prop.gcnArchName, lanes, (void*)slab, steps);
    struct { size_t chunk; size_t stride; size_t first; } v[] = {
        { (size_t)8 << 30, 1, 0 }, { (size_t)8 << 30, 1, (size_t)64 << 30 }, { (size_t)8 << 30, 1, (size_t)120 << 30 },
        { (size_t)1 << 30, 2, 0 }, { (size_t)1 << 30, 4, 0 }, { (size_t)1 << 30, 8, 0 }, { (size_t)1 << 30, 16, 0 },
        { (size_t)128 << 20, 2, 0 }, { (size_t)128 << 20, 4, 0 }, { (size_t)128 << 20, 8, 0 }, { (size_t)128 << 20, 16, 0 },
        { (size_t)2 << 20, 2, 0 }, { (size_t)2 << 20, 4, 0 }, { (size_t)2 << 20, 8, 0 }, { (size_t)2 << 20, 16, 0 },
        { (size_t)32 << 10, 2, 0 }, { (size_t)32 << 10, 4, 0 }, { (size_t)32 << 10, 16, 0 },
        { (size_t)8 << 30, 1, 0 },
    };
    for (auto& t : v) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        chase<<<groups, 64>>>(slab + t.first, region, t.chunk, t.stride, 200, sink);
        (void)hipEventRecord(a);
        chase<<<groups, 64>>>(slab + t.first, region, t.chunk, t.stride, steps, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("chunks of %8zu KiB, every %2zu-th (8 GiB spread over %3zu GiB from +%3zu GiB): %8.2f ms  %6.2f G steps/s\n", t.chunk >> 10, t.stride,
               (size_t)8 * t.stride, t.first >> 30, ms, (double)lanes * steps / ms / 1e6);
    }
    return 0;
}
