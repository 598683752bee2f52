// Microbenchmark (not part of the product): what does a per-lane DYNAMIC index into a small register array cost on gfx950?
// The lane decoder's input window would live in registers if 16 bytes at a per-lane byte position could be taken out of it
// cheaply.  Variants, per loop iteration (one "view" = 5 consecutive dwords at dword index k, then a byte rotation):
//   0  empty loop (LCG + accumulate only)
//   1  v_cndmask tree over 20 dwords, k in 0..15 (4 levels, 31 selects)     -- 64-byte sector window + 16-byte tail
//   2  v_cndmask tree over 12 dwords, k in 0..7  (3 levels, 19 selects)     -- 32-byte piece window + 16-byte tail
//   3  as 1 plus the window switch (20 conditional moves)
//   4  as 2 plus the window switch (12 conditional moves)
//   5  the same view from LDS (5 ds_read_b32 at a per-lane row, dword-interleaved layout) -- the staging ring of generation 3
// Reports SIMD-cycles per iteration per wavefront at 1 / 3 / 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// (written with inline assembly: given the plain C++ select tree the compiler recognises a dynamically indexed array and moves it to scratch memory)
__device__ __forceinline__ uint32_t sel(uint64_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
// four selects under one condition as VOP2 (implicit VCC): is the e32 form cheaper than the e64 form with an SGPR-pair mask?
__device__ __forceinline__ void sel4_vcc(uint64_t m, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                         uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3)
{
    asm("s_mov_b64 vcc, %[m]\n\tv_cndmask_b32_e32 %[r0], %[b0], %[a0], vcc\n\tv_cndmask_b32_e32 %[r1], %[b1], %[a1], vcc\n\tv_cndmask_b32_e32 %[r2], %[b2], %[a2], vcc\n\tv_cndmask_b32_e32 %[r3], %[b3], %[a3], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
        : [m] "s"(m), [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [a3] "v"(a3), [b3] "v"(b3) : "vcc");
}
__device__ __forceinline__ void view_tree12_vcc(const uint32_t (&w)[20], uint32_t k, uint32_t (&v)[5])
{
    const uint64_t c2 = __builtin_amdgcn_ballot_w64((k & 4u) != 0), c1 = __builtin_amdgcn_ballot_w64((k & 2u) != 0), c0 = __builtin_amdgcn_ballot_w64((k & 1u) != 0);
    uint32_t b[8], c[8], d[8];
    sel4_vcc(c2, b[0], b[1], b[2], b[3], w[4], w[0], w[5], w[1], w[6], w[2], w[7], w[3]);
    sel4_vcc(c2, b[4], b[5], b[6], b[7], w[8], w[4], w[9], w[5], w[10], w[6], w[11], w[7]);
    sel4_vcc(c1, c[0], c[1], c[2], c[3], b[2], b[0], b[3], b[1], b[4], b[2], b[5], b[3]);
    sel4_vcc(c1, c[4], c[5], c[6], c[7], b[6], b[4], b[7], b[5], b[7], b[6], b[7], b[7]);
    sel4_vcc(c0, d[0], d[1], d[2], d[3], c[1], c[0], c[2], c[1], c[3], c[2], c[4], c[3]);
    sel4_vcc(c0, d[4], d[5], d[6], d[7], c[5], c[4], c[5], c[5], c[5], c[5], c[5], c[5]);
    for (int i = 0; i < 5; i++) v[i] = d[i];
}
__device__ __forceinline__ void view_tree20(const uint32_t (&w)[20], uint32_t k, uint32_t (&v)[5])
{
    const uint64_t c3 = __builtin_amdgcn_ballot_w64((k & 8u) != 0), c2 = __builtin_amdgcn_ballot_w64((k & 4u) != 0), c1 = __builtin_amdgcn_ballot_w64((k & 2u) != 0), c0 = __builtin_amdgcn_ballot_w64((k & 1u) != 0);
    uint32_t a[12], b[8], c[6];
#pragma unroll
    for (int i = 0; i < 12; i++) a[i] = sel(c3, w[i + 8], w[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = sel(c2, a[i + 4], a[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = sel(c1, b[i + 2], b[i]);
#pragma unroll
    for (int i = 0; i < 5; i++) v[i] = sel(c0, c[i + 1], c[i]);
}
__device__ __forceinline__ void view_tree12(const uint32_t (&w)[20], uint32_t k, uint32_t (&v)[5])
{
    const uint64_t c2 = __builtin_amdgcn_ballot_w64((k & 4u) != 0), c1 = __builtin_amdgcn_ballot_w64((k & 2u) != 0), c0 = __builtin_amdgcn_ballot_w64((k & 1u) != 0);
    uint32_t b[8], c[6];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = sel(c2, w[i + 4], w[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = sel(c1, b[i + 2], b[i]);
#pragma unroll
    for (int i = 0; i < 5; i++) v[i] = sel(c0, c[i + 1], c[i]);
}

template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t* out, const uint32_t* in, int iters)
{
    __shared__ uint32_t lds[64 * 32];
    uint32_t w[20], nx[20];
    for (int j = 0; j < 20; j++) { w[j] = in[threadIdx.x * 20 + j]; nx[j] = in[(threadIdx.x ^ 1) * 20 + j]; }
    for (int j = 0; j < 32; j++) lds[j * 64 + threadIdx.x] = w[j % 20] + j;
    uint32_t s = threadIdx.x * 2654435761u + 1u, acc = 0, d = threadIdx.x & 63;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        d += (s >> 28);                                            // the cursor advances 0..15 bytes
        uint32_t v[5] = { 0, 0, 0, 0, 0 };
        if (KIND == 1 || KIND == 3) {
            const bool cross = d >= 64;
            const uint64_t cm = __builtin_amdgcn_ballot_w64(cross);
            if (KIND == 3) {
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = sel(cm, w[16 + j], w[j]);
#pragma unroll
                for (int j = 0; j < 16; j++) w[4 + j] = sel(cm, nx[j], w[4 + j]);
            }
            d = cross ? d - 64 : d;
            view_tree20(w, d >> 2, v);
        } else if (KIND == 2 || KIND == 4) {
            const bool cross = d >= 32;
            const uint64_t cm = __builtin_amdgcn_ballot_w64(cross);
            if (KIND == 4) {
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = sel(cm, w[8 + j], w[j]);
#pragma unroll
                for (int j = 0; j < 8; j++) w[4 + j] = sel(cm, nx[j], w[4 + j]);
            }
            d = cross ? d - 32 : d;
            view_tree12(w, d >> 2, v);
        } else if (KIND == 6) {
            const bool cross = d >= 32;
            d = cross ? d - 32 : d;
            view_tree12_vcc(w, d >> 2, v);
        } else if (KIND == 5) {
            d &= 63;
            const uint32_t row = d >> 2;
#pragma unroll
            for (int j = 0; j < 5; j++) v[j] = lds[((row + j) & 31) * 64 + threadIdx.x];
        } else {
            d &= 63;
            v[0] = d; v[1] = s; v[2] = acc; v[3] = d ^ s; v[4] = s >> 3;
        }
        const uint32_t sel = __builtin_amdgcn_alignbyte(0x07060504u, 0x03020100u, d & 3u);
        acc += __builtin_amdgcn_perm(v[1], v[0], sel) ^ __builtin_amdgcn_perm(v[2], v[1], sel) ^ __builtin_amdgcn_perm(v[3], v[2], sel) ^ __builtin_amdgcn_perm(v[4], v[3], sel);
        if (KIND == 3 || KIND == 4) { nx[0] ^= acc; nx[9] += acc; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + w[3] + nx[5];
}

template <int KIND>
void run(const char* name, uint32_t* d, const uint32_t* in, int cus, float* base)
{
    const int iters = 20000;
    int col = 0;
    for (int w : { 1, 3, 4 }) {
        const int grid = cus * 4 * w;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<KIND><<<grid, 64>>>(d, in, 100);
        (void)hipEventRecord(a);
        k<KIND><<<grid, 64>>>(d, in, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * w);
        if (KIND == 0) base[col] = (float)cyc;
        printf("%-52s waves/SIMD %d: %8.3f ms  %7.1f SIMD-cycles per iteration per wavefront (%+7.1f vs empty loop)\n", name, w, ms, cyc, cyc - base[col]);
        col++;
    }
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t *d, *in; (void)hipMalloc(&d, (size_t)cus * 4 * 8 * 64 * 4); (void)hipMalloc(&in, 64 * 20 * 4); (void)hipMemset(in, 3, 64 * 20 * 4);
    float base[3] = { 0, 0, 0 };
    run<0>("empty loop", d, in, cus, base);
    run<1>("tree over 20 dwords (31 selects)", d, in, cus, base);
    run<2>("tree over 12 dwords (19 selects)", d, in, cus, base);
    run<3>("tree over 20 dwords + window switch (20 moves)", d, in, cus, base);
    run<4>("tree over 12 dwords + window switch (12 moves)", d, in, cus, base);
    run<5>("LDS rows (5 ds_read_b32)", d, in, cus, base);
    run<6>("tree over 12 dwords as VOP2 + VCC (24 selects)", d, in, cus, base);
    return 0;
}
