// Microbenchmark (not part of the product): the fast lane encoder's memory behaviour without the encoder.  262 144 lanes (16 wavefronts per CU),
// each with a private region of a slab (the encoder: a 32 KiB hash table per lane, 8.6 GB in all), each performing a DEPENDENT chain of random
// 4-byte read-modify-writes inside its region -- one 64-byte sector in, one sector out per step, nothing stays in a cache.  The question
// (the encoder's two rates): does the rate depend on the slab's FOOTPRINT beyond all cache sizes, i.e. on address translation?
//   ./a.out [steps per lane]      prints G steps/s for footprints of 0.25 .. 32 GiB (region per lane 1 KiB .. 128 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ void __launch_bounds__(64) chase(uint32_t* slab, size_t region_words, int steps, uint32_t* sink)
{
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t* const p = slab + lane * region_words;
    uint32_t s = (uint32_t)lane * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < steps; i++) {
        s = s * 1664525u + 1013904223u;
        const size_t off = (size_t)((s >> 4) % (uint32_t)region_words);
        const uint32_t v = p[off];
        p[off] = v + (uint32_t)i;
        s ^= v * 0x9E3779B9u;                                        // the next address depends on what was read
        acc += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int groups = prop.multiProcessorCount * 16;
    const size_t lanes = (size_t)groups * 64;
    const size_t max_bytes = (size_t)32 << 30;
    uint32_t *slab = nullptr, *sink = nullptr;
    if (hipMalloc(&slab, max_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(slab, 1, max_bytes);
    (void)hipDeviceSynchronize();
    printf("%s: %zu lanes (16 wavefronts per CU), %d dependent random read-modify-writes per lane, slab at %p\n", prop.gcnArchName, lanes, steps, (void*)slab);
    const double gib[] = { 0.25, 1, 2, 4, 8, 16, 32, 8, 1 };
    for (double g : gib) {
        const size_t region_words = (size_t)(g * (double)((size_t)1 << 30) / (double)lanes) / 4;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        chase<<<groups, 64>>>(slab, region_words, 200, sink);
        (void)hipEventRecord(a);
        chase<<<groups, 64>>>(slab, region_words, steps, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("footprint %6.2f GiB (region per lane %7zu bytes): %8.2f ms  %6.2f G steps/s  = %5.2f TB/s of sector traffic (64 B in + 64 B out per step)\n",
               g, region_words * 4, ms, (double)lanes * steps / ms / 1e6, (double)lanes * steps * 128 / ms / 1e9);
    }
    // Does the rate depend on WHERE in device memory the slab lies?  Seven more 32 GiB allocations (the first one stays), the 8 GiB test at the
    // start and in the middle of each.
    if (argc > 2) {
        uint32_t* more[8] = { slab };
        int n = 1;
        for (; n < 8; n++) if (hipMalloc(&more[n], max_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        const size_t region_words = ((size_t)8 << 30) / lanes / 4;
        for (int k = 0; k < n; k++) {
            if (k) (void)hipMemset(more[k], 1, max_bytes);
            for (int half = 0; half < 2; half++) {
                uint32_t* const base = more[k] + (half ? ((size_t)16 << 30) / 4 : 0);
                hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                chase<<<groups, 64>>>(base, region_words, 200, sink);
                (void)hipEventRecord(a);
                chase<<<groups, 64>>>(base, region_words, steps, sink);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
                printf("allocation %d (%p) + %2d GiB: 8 GiB footprint %8.2f ms  %6.2f G steps/s\n", k, (void*)more[k], half * 16, ms, (double)lanes * steps / ms / 1e6);
            }
        }
    }
    // Is it the ALIGNMENT of the slab's virtual address (the page tables describe larger contiguous fragments when it is high)?  A fresh 10 GiB
    // allocation per trial, the 8 GiB test at the first address inside it that is aligned to 2 MiB .. 2 GiB but NOT to twice that.
    if (argc > 3) {
        for (int rep = 0; rep < 2; rep++)
            for (int lg = 21; lg <= 31; lg += 2) {
                uint32_t* raw = nullptr;
                const size_t want = ((size_t)8 << 30) + ((size_t)2 << lg) + ((size_t)4 << 20);
                if (hipMalloc(&raw, want) != hipSuccess) { (void)hipGetLastError(); printf("allocation failed\n"); continue; }
                (void)hipMemset(raw, 1, want);
                const uintptr_t a = (uintptr_t)1 << lg;
                uintptr_t p0 = ((uintptr_t)raw + a - 1) & ~(a - 1);
                if ((p0 & a) == 0 && lg < 31) p0 += a;                 // aligned to 2^lg, not to 2^(lg+1)
                const size_t region_words = ((size_t)8 << 30) / lanes / 4;
                hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
                chase<<<groups, 64>>>((uint32_t*)p0, region_words, 200, sink);
                (void)hipEventRecord(ea);
                chase<<<groups, 64>>>((uint32_t*)p0, region_words, steps, sink);
                (void)hipEventRecord(eb); (void)hipEventSynchronize(eb);
                float ms = 0; (void)hipEventElapsedTime(&ms, ea, eb);
                printf("hipMalloc -> %p, slab at %p (aligned to 2^%d): 8 GiB footprint %8.2f ms  %6.2f G steps/s\n", (void*)raw, (void*)p0, lg, ms, (double)lanes * steps / ms / 1e6);
                (void)hipFree(raw);
            }
    }
    return 0;
}
