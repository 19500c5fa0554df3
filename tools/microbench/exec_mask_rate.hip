// Does a VALU instruction cost less when part of the wave is switched off in EXEC?  (A wave64 instruction goes through the
// 16-lane SIMD in four passes; if passes whose lanes are all inactive are skipped, exec-masked regions over spatially
// coherent pixels are cheaper than predicated code that keeps every lane busy.)  8 waves per SIMD, independent FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define REP8(op) op " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" \
                 op " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n"
#define REP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float a, int iters, unsigned long long mask)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    if (on) {   // the loop runs under the partial EXEC mask
        for (int i = 0; i < iters; ++i) {
            if (OP == 0) asm volatile(REP8("v_fma_f32") : R8 : "v"(a));
            else asm volatile(REP8_1("v_exp_f32") : R8 : "v"(a));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int OP> void run(const char* name, unsigned long long mask)
{
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000, blocks = 2048;
    k<OP><<<blocks, 256>>>(d, 1.0001f, 10, mask);
    (void)hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 1.0001f, iters, mask); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;
    printf("%-10s exec %016llx : %.2f ns per wave64 instruction per SIMD\n", name, mask, ms * 1e6 / per_simd);
    (void)hipFree(d);
}
int main()
{
    const unsigned long long masks[] = {~0ull, 0x00000000FFFFFFFFull, 0x000000000000FFFFull, 0x00000000000000FFull, 0x0000FFFF0000FFFFull,
                                        0x5555555555555555ull, 0x00FF00FF00FF00FFull, 0x0000000000000001ull, 0xFFFF00000000FFFFull};
    for (unsigned long long m : masks) run<0>("v_fma_f32", m);
    for (unsigned long long m : masks) run<1>("v_exp_f32", m);
}
