// Microbenchmark (not part of the product), fourth part of tools/microbench_random_sectors.hip: the LZ4HC lane kernel's memory behaviour -- 262 144
// lanes, each a dependent chain of random 4-byte READS inside a private 256 KiB table (64 GiB in all) -- with the tables in one contiguous 64 GiB
// piece of a 192 GiB allocation or as 1 GiB chunks spread over 128 / 192 GiB.  Does the spread matter for reads as it does for read-modify-writes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <bool WRITE>
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
        if (WRITE) p[off] = v + (uint32_t)i;
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
    const size_t lanes = (size_t)groups * 64, total = (size_t)192 << 30;
    uint8_t* slab = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&slab, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(slab, 1, total);
    (void)hipDeviceSynchronize();
    printf("%s: %zu lanes, %d dependent random accesses per lane, 192 GiB allocation at %p\n", prop.gcnArchName, lanes, steps, (void*)slab);
    struct { size_t region; size_t chunk; size_t stride; } v[] = {
        { 262144, (size_t)64 << 30, 1 }, { 262144, (size_t)1 << 30, 2 }, { 262144, (size_t)1 << 30, 3 }, { 262144, (size_t)64 << 20, 3 },
        { 32768, (size_t)8 << 30, 1 }, { 32768, (size_t)128 << 20, 8 }, { 32768, (size_t)128 << 20, 24 },
    };
    for (auto& t : v)
        for (int w = 0; w < 2; w++) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            if (w) chase<true><<<groups, 64>>>(slab, t.region, t.chunk, t.stride, 200, sink); else chase<false><<<groups, 64>>>(slab, t.region, t.chunk, t.stride, 200, sink);
            (void)hipEventRecord(a);
            if (w) chase<true><<<groups, 64>>>(slab, t.region, t.chunk, t.stride, steps, sink); else chase<false><<<groups, 64>>>(slab, t.region, t.chunk, t.stride, steps, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            const double gib = (double)(lanes * t.region) / 1073741824.0;
            printf("%-18s region %6zu KiB (%4.0f GiB in all), chunks of %8zu KiB, every %2zu-th (spread over %4.0f GiB): %8.2f ms  %6.2f G steps/s\n", w ? "read-modify-write" : "read only",
                   t.region >> 10, gib, t.chunk >> 10, t.stride, gib * (double)t.stride, ms, (double)lanes * steps / ms / 1e6);
        }
    return 0;
}
