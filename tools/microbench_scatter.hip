// Microbenchmark (not part of the product): cost of lane-private streaming accesses on gfx950.
// Every lane walks its own 64 KiB region (stride 64 KiB between lanes, like one LZ4 block per lane),
// reading or writing `W` bytes per step.  Reports GB/s and lane-requests per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int W, bool STORE, int ACTIVE>
__global__ void __launch_bounds__(64) walk(uint8_t* base, int steps, int stride, uint64_t* sink)
{
    const int64_t lane_global = (int64_t)blockIdx.x * 64 + threadIdx.x;
    uint8_t* p = base + lane_global * 65536;
    if ((threadIdx.x % 64) >= ACTIVE) return;
    uint64_t acc = 0;
    for (int i = 0; i < steps; i++) {
        uint8_t* q = p + (int64_t)i * stride;
        if (STORE) {
            if (W == 16) { uint4 v = make_uint4(i, i, i, i); __builtin_memcpy(q, &v, 16); }
            if (W == 8) { uint64_t v = i; __builtin_memcpy(q, &v, 8); }
            if (W == 4) { uint32_t v = i; __builtin_memcpy(q, &v, 4); }
        } else {
            if (W == 16) { uint4 v; __builtin_memcpy(&v, q, 16); acc += v.x + v.w; }
            if (W == 8) { uint64_t v; __builtin_memcpy(&v, q, 8); acc += v; }
            if (W == 4) { uint32_t v; __builtin_memcpy(&v, q, 4); acc += v; }
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
template <int W, bool STORE, int ACTIVE>
void run(uint8_t* buf, uint64_t* sink, int waves, int steps, int stride, const char* name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((walk<W, STORE, ACTIVE>), dim3(waves), dim3(64), 0, 0, buf, 8, stride, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((walk<W, STORE, ACTIVE>), dim3(waves), dim3(64), 0, 0, buf, steps, stride, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double reqs = (double)waves * ACTIVE * steps;
    printf("%-28s waves=%6d active=%2d stride=%4d: %8.3f ms  %8.1f GB/s  %6.3f lane-req/ns  (%.2f cycles/lane-req/CU @2.4GHz,256CU)\n",
           name, waves, ACTIVE, stride, ms, reqs * W / ms / 1e6, reqs / ms / 1e6, 256.0 * 2.4e6 * ms / reqs);
}
int main()
{
    const int max_waves = 8192;
    uint8_t* buf; uint64_t* sink;
    hipMalloc(&buf, (size_t)max_waves * 64 * 65536); hipMalloc(&sink, 8);
    hipMemset(buf, 1, (size_t)max_waves * 64 * 65536);
    for (int waves : {1024, 2560, 8192}) {
        run<16, false, 64>(buf, sink, waves, 2048, 16, "load16 sequential");
        run<8, false, 64>(buf, sink, waves, 2048, 8, "load8 sequential");
        run<4, false, 64>(buf, sink, waves, 2048, 4, "load4 sequential");
        run<16, false, 64>(buf, sink, waves, 512, 128, "load16 line-stride");
        run<16, true, 64>(buf, sink, waves, 2048, 16, "store16 sequential");
        run<8, true, 64>(buf, sink, waves, 2048, 8, "store8 sequential");
        run<16, true, 64>(buf, sink, waves, 512, 128, "store16 line-stride");
        run<16, false, 8>(buf, sink, waves, 2048, 16, "load16 seq, 8 lanes");
        run<16, true, 8>(buf, sink, waves, 2048, 16, "store16 seq, 8 lanes");
    }
    return 0;
}
