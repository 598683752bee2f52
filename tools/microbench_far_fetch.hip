// Microbenchmark (not part of the product): what does a lane decoder's FAR-MATCH FETCH cost the memory system on gfx950?
// Every lane owns a 64 KiB region (one LZ4 block per lane, regions 64 KiB apart) and reads 16 bytes at pseudo-random
// positions inside a window of `window` bytes that slides through the region -- the access pattern of far matches.
// Variants: cache-policy bits on the load (plain / nt / sc1 / sc0 sc1), and a second 16-byte load in the SAME 128-byte
// line but the OTHER 64-byte half (is the half that comes along free, i.e. does a miss fill 128 bytes?).
// Reports lane-requests per second; run under rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum /
// TCC_MISS_sum for the bytes per request.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ u32x4 load16(const uint8_t* p)
{
    u32x4 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// MODE 0: one random 16-byte load per step; MODE 1: plus the same offset in the other 64-byte half of the 128-byte line;
// MODE 2: plus a load in a DIFFERENT random line (control: two independent misses)
template <int POLICY, int MODE>
__global__ void __launch_bounds__(64) far_fetch(const uint8_t* base, int steps, int window, uint64_t* sink)
{
    const int64_t lane_global = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const uint8_t* p = base + lane_global * 65536;
    uint32_t s = (uint32_t)lane_global * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < steps; i += 4) {
        u32x4 v[4] = {}, w[4] = {};
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            s = s * 1664525u + 1013904223u;
            const int pos = (int)(((uint64_t)(i + k) * (65536 - window)) / steps);        // the window slides forward
            const int off = (int)((s >> 8) % (uint32_t)(window - 16));
            const uint8_t* q = p + pos + off;
            v[k] = load16<POLICY>(q);
            if (MODE == 1) w[k] = load16<POLICY>((const uint8_t*)((uint64_t)q ^ 64));
            if (MODE == 2) { s = s * 1664525u + 1013904223u; w[k] = load16<POLICY>(p + pos + (int)((s >> 8) % (uint32_t)(window - 16))); }
        }
        // (the wait names the destination registers: the compiler must not read them before it)
        if (MODE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) :: "memory");
        else      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
        #pragma unroll
        for (int k = 0; k < 4; k++) { acc += v[k].x + v[k].w; if (MODE) acc += w[k].y; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int POLICY, int MODE>
void run(const uint8_t* buf, uint64_t* sink, int waves, int steps, int window, const char* name)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((far_fetch<POLICY, MODE>), dim3(waves), dim3(64), 0, 0, buf, 8, window, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((far_fetch<POLICY, MODE>), dim3(waves), dim3(64), 0, 0, buf, steps, window, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double steps_total = (double)waves * 64 * steps;
    printf("%-34s window=%5d waves=%5d: %8.3f ms  %7.2f G steps/s (%d load(s) per step)\n", name, window, waves, ms, steps_total / ms / 1e6, MODE ? 2 : 1);
    fflush(stdout);
}

int main()
{
    const int waves = 3072;                                   // 12 per CU, as the lane decoder
    uint8_t* buf; uint64_t* sink;
    (void)hipMalloc(&buf, (size_t)waves * 64 * 65536 + 4096); (void)hipMalloc(&sink, 8);   // (+ slack: "^ 64" may step past the last region)
    (void)hipMemset(buf, 1, (size_t)waves * 64 * 65536 + 4096);
    for (int window : {1024, 32768}) {
        run<0, 0>(buf, sink, waves, 2048, window, "plain");
        run<1, 0>(buf, sink, waves, 2048, window, "nt");
        run<2, 0>(buf, sink, waves, 2048, window, "sc1");
        run<3, 0>(buf, sink, waves, 2048, window, "sc0 sc1");
        run<4, 0>(buf, sink, waves, 2048, window, "sc0 sc1 nt");
        run<0, 1>(buf, sink, waves, 2048, window, "plain + other half of the line");
        run<0, 2>(buf, sink, waves, 2048, window, "plain + another random line");
        run<1, 1>(buf, sink, waves, 2048, window, "nt + other half of the line");
    }
    return 0;
}
