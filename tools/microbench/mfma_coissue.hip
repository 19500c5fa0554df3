// Does the matrix pipe run beside the VALU on gfx950?  A stream of independent VGPR-only v_fma_f32 (24 per loop body) with
// 0 / 1 / 2 / 4 / 8 v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles in the matrix pipe each) mixed in, 8 waves per SIMD.
// If the MFMAs only cost their issue slot, the matrix pipe is a free cross-lane adder for kernels whose VALU is saturated
// (the backward blend's per-entry reduction: DESIGN.md section 4).   make -C tools/microbench && bin/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define R8 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define FMA8 "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n" \
             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
template <int NM, bool CHAIN>
__global__ void __launch_bounds__(256) k(float* out, float a, int iters)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v4f d0 = {0, 0, 0, 0}, d1 = d0;
    const float one = 1.0f, ma = x0;
    for (int i = 0; i < iters; ++i) {
        // NM MFMAs spread over the body's three groups of eight FMAs; CHAIN: all accumulate into d0 (a dependent chain,
        // as the reduction would), else alternate d0 / d1
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            asm volatile(FMA8 : R8 : "v"(a));
#pragma unroll
            for (int m = 0; m < (NM + 2 - g) / 3; ++m) {
                if (CHAIN || ((m + g) & 1) == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d0) : "v"(ma), "v"(one));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d1) : "v"(ma), "v"(one));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + d0.x + d0.y + d1.z + d1.w;
}
template <int NM, bool CHAIN> void run()
{
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 8000, blocks = 2048;
    k<NM, CHAIN><<<blocks, 256>>>(d, 1.0001f, 10);
    (void)hipEventRecord(a); k<NM, CHAIN><<<blocks, 256>>>(d, 1.0001f, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bodies_per_simd = (double)blocks * 4 * iters / 1024.0;
    printf("24 v_fma + %d mfma 16x16x4 f32 (%s): %8.3f ms -> %6.1f ns per body per SIMD (24 FMAs alone: ~31 ns)\n", NM,
           CHAIN ? "one accumulator" : "two accumulators", ms, ms * 1e6 / bodies_per_simd);
    (void)hipFree(d);
}
int main()
{
    run<0, false>(); run<1, false>(); run<2, false>(); run<4, false>(); run<8, false>();
    run<2, true>(); run<4, true>(); run<8, true>();
}
