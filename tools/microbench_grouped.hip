// Microbenchmark (not part of the product): does the vector-memory path of gfx950 charge a wavefront
// instruction per LANE or per 64-byte LINE?  G lanes cooperate on one stream: group g = lane / G walks its
// own 64 KiB-strided region, lane sub = lane % G accesses bytes [16*sub, 16*sub+16) of the group's next
// G*16-byte piece.  G = 1 is the lane-per-block pattern (64 lines per instruction), G = 4 touches 16 full
// 64-byte lines per instruction with the same number of bytes and lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int G, bool STORE>
__global__ void __launch_bounds__(64) walk(uint8_t* base, int steps, uint64_t* sink)
{
    const int lane = threadIdx.x, grp = lane / G, sub = lane % G;
    // every group owns a 64 KiB region; a wave owns 64 regions (only 64/G of them are used when G > 1)
    uint8_t* p = base + ((int64_t)blockIdx.x * 64 + grp) * 65536 + sub * 16;
    uint32_t acc = 0;
    for (int i = 0; i < steps; i++) {
        uint8_t* q = p + (int64_t)i * (16 * G);
        if (STORE) { uint4 v = make_uint4(i, i, i, lane); __builtin_memcpy(q, &v, 16); }
        else       { uint4 v; __builtin_memcpy(&v, q, 16); acc += v.x + v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int G, bool STORE>
void run(uint8_t* buf, uint64_t* sink, int waves, int bytes_per_group, const char* name)
{
    const int steps = bytes_per_group / (16 * G);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((walk<G, STORE>), dim3(waves), dim3(64), 0, 0, buf, 4, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<G, STORE>), dim3(waves), dim3(64), 0, 0, buf, steps, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instrs = (double)waves * steps, bytes = instrs * 64 * 16;
    printf("%-10s G=%d waves=%5d: %8.3f ms %8.1f GB/s  %7.1f CU-cycles per wave-instruction (%d lines each)\n",
           name, G, waves, ms, bytes / ms / 1e6, 256.0 * 2.4e6 * ms / instrs, 64 / G);
}
int main()
{
    const int max_waves = 8192;
    uint8_t* buf; uint64_t* sink;
    (void)hipMalloc(&buf, (size_t)max_waves * 64 * 65536); (void)hipMalloc(&sink, 8);
    (void)hipMemset(buf, 1, (size_t)max_waves * 64 * 65536);
    for (int waves : {2560, 5120}) {
        run<1, false>(buf, sink, waves, 32768, "load16");
        run<2, false>(buf, sink, waves, 32768, "load16");
        run<4, false>(buf, sink, waves, 32768, "load16");
        run<8, false>(buf, sink, waves, 32768, "load16");
        run<1, true>(buf, sink, waves, 32768, "store16");
        run<2, true>(buf, sink, waves, 32768, "store16");
        run<4, true>(buf, sink, waves, 32768, "store16");
        run<8, true>(buf, sink, waves, 32768, "store16");
    }
    return 0;
}
