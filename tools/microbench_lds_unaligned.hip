// Microbenchmark (not part of the product): cost of lane-private LDS accesses on gfx950 at byte-granular
// (unaligned) addresses versus aligned ones, for the two candidate layouts of a per-lane byte ring:
//   interleaved : lane i's qword k lives at (k * 64 + i) * 8   -- always bank-conflict free, aligned only
//   lane-major  : lane i owns STRIDE contiguous bytes           -- unaligned access possible, random conflicts
// Each lane performs `iters` dependent {write W bytes at a, read W bytes at a'} pairs at pseudo-random offsets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct __attribute__((packed)) P16 { uint32_t w[4]; };
struct __attribute__((packed)) P8 { uint64_t v; };

template <int MODE, int W, int STRIDE>
__global__ void __launch_bounds__(64) ring(int iters, uint64_t* sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    uint32_t a = lane * 7 + blockIdx.x, acc = 0;
    uint8_t* mine = lds + lane * STRIDE + 16;
    for (int i = 0; i < iters; i++) {
        a = a * 1664525u + 1013904223u + acc;
        uint32_t wa = (a >> 8) & 127, ra = (a >> 16) & 127;
        if (MODE == 0) {           // interleaved qwords, aligned 8-byte accesses (2 per 16 bytes)
            uint64_t* q = (uint64_t*)lds;
            const int kw = wa >> 3, kr = ra >> 3;
            q[(kw & 15) * 64 + lane] = a; q[((kw + 1) & 15) * 64 + lane] = a + 1;
            if (W == 16) q[((kw + 2) & 15) * 64 + lane] = a + 2;
            uint64_t x = q[(kr & 15) * 64 + lane] + q[((kr + 1) & 15) * 64 + lane];
            if (W == 16) x += q[((kr + 2) & 15) * 64 + lane];
            acc += (uint32_t)x;
        } else {
            if (MODE == 1) { wa &= ~(W - 1); ra &= ~(W - 1); }   // lane-major, aligned
            if (W == 16) {
                P16 v = { { a, a + 1, a + 2, a + 3 } };
                *(P16*)(mine + wa) = v;
                P16 r = *(const P16*)(mine + ra);
                acc += r.w[0] + r.w[3];
            } else {
                P8 v = { a };
                *(P8*)(mine + wa) = v;
                P8 r = *(const P8*)(mine + ra);
                acc += (uint32_t)r.v + (uint32_t)(r.v >> 32);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int W, int STRIDE>
void run(uint64_t* sink, int waves, int iters, const char* name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = MODE == 0 ? 64 * 128 : 64 * STRIDE + 64;
    hipLaunchKernelGGL((ring<MODE, W, STRIDE>), dim3(waves), dim3(64), lds, 0, 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((ring<MODE, W, STRIDE>), dim3(waves), dim3(64), lds, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-iterations per CU per cycle -> cycles per wave-iteration per CU
    const double wave_iters = (double)waves * iters;
    printf("%-44s W=%2d stride=%3d lds=%5zu waves=%5d: %8.3f ms  %7.1f CU-cycles per wave-iteration (write+read)\n",
           name, W, STRIDE, lds, waves, ms, 256.0 * 2.4e6 * ms / wave_iters);
}

int main()
{
    uint64_t* sink; hipMalloc(&sink, 8);
    for (int waves : {256 * 8, 256 * 16}) {
        run<0, 8, 0>(sink, waves, 20000, "interleaved qwords, aligned (2w+2r b64)");
        run<0, 16, 0>(sink, waves, 20000, "interleaved qwords, aligned (3w+3r b64)");
        run<1, 8, 176>(sink, waves, 20000, "lane-major aligned b64");
        run<2, 8, 176>(sink, waves, 20000, "lane-major UNALIGNED b64");
        run<1, 16, 176>(sink, waves, 20000, "lane-major aligned b128");
        run<2, 16, 176>(sink, waves, 20000, "lane-major UNALIGNED b128");
        run<2, 16, 180>(sink, waves, 20000, "lane-major UNALIGNED b128");
        run<2, 16, 196>(sink, waves, 20000, "lane-major UNALIGNED b128");
        run<2, 8, 180>(sink, waves, 20000, "lane-major UNALIGNED b64");
    }
    return 0;
}
