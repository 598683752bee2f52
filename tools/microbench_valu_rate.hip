// Issue rate of the integer VALU instructions the lane decoder is made of, per SIMD, at 1 / 2 / 4 wavefronts per SIMD.
// Each wavefront runs ITER x 32 independent instructions of one kind (8 accumulators); reports SIMD-cycles per
// wavefront-instruction assuming 2.4 GHz.   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu_rate.hip -o tools/microbench_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint32_t s)
{
    uint32_t a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j;
    uint32_t b = s ^ threadIdx.x, c = s + 77u;
    uint64_t m64 = 0x5555AAAA3333CCCCull ^ s;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (KIND == 2) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));
                if (KIND == 4) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (KIND == 5) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 7) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc");
                if (KIND == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(uint64_t*)&a[j & 6]) : "v"(*(uint64_t*)&a[(j & 6) ^ 2]));
                if (KIND == 9) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[0]));          // dependent chain
                if (KIND == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(m64));
                if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 13) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[j]));
                if (KIND == 14) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[j]));
                if (KIND == 15) asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m64) : "v"(a[j]), "v"(b));
                if (KIND == 16) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(b) : "vcc");
                if (KIND == 17) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (KIND == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (KIND == 19) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[j]) : "v"(b), "v"(c));       // no read of dst
            }
        }
    }
    uint32_t x = 0;
    for (int j = 0; j < 8; j++) x ^= a[j];
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int KIND>
void run(const char* name, uint32_t* d, int cus)
{
    const int iters = 20000;
    for (int w : { 1, 3, 8 }) {
        const int grid = cus * 4 * w;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<KIND><<<grid, 64>>>(d, 100, 1);
        hipEventRecord(a);
        k<KIND><<<grid, 64>>>(d, iters, 1);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * w);
        printf("%-18s waves/SIMD %d: %.3f ms  %.2f SIMD-cycles per wavefront-instruction (at 2.4 GHz)\n", name, w, ms, cyc);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    uint32_t* d; hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * 4);
    run<0>("v_add_u32", d, cus); run<1>("v_perm_b32", d, cus); run<2>("v_and_or_b32", d, cus); run<3>("v_cndmask_b32", d, cus);
    run<4>("v_alignbyte_b32", d, cus); run<5>("v_lshl_add_u32", d, cus); run<6>("v_mul_lo_u32", d, cus); run<7>("v_cmp_lt_u32", d, cus);
    run<8>("v_lshl_add_u64", d, cus); run<9>("dependent v_add", d, cus);
    run<10>("v_cndmask_e64 sgpr", d, cus); run<11>("v_mov_b32", d, cus); run<12>("v_and_b32", d, cus); run<13>("v_lshlrev_b32", d, cus);
    run<14>("v_bfe_u32", d, cus); run<15>("v_cmp_e64 ->sgpr", d, cus); run<16>("v_cmp+v_cndmask vcc (x2)", d, cus); run<17>("v_add3_u32", d, cus);
    run<18>("v_sub_u32", d, cus); run<19>("v_add_u32 d=b+c", d, cus);
    return 0;
}
