// Issue rate of single VALU instructions on gfx950, per wave64 per SIMD (the figures DESIGN.md section 4 prices the blend
// kernels with).  Every sequence is inline asm on independent registers, so neither the compiler's SLP vectoriser nor
// its scheduler changes what is measured.  2048 workgroups x 256 threads = 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R8 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define REP8(op) op " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" \
                 op " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n"
#define REP8_2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                   op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define REP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, int iters)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f q0 = {x0, x1}, q1 = {x2, x3}, q2 = {x4, x5}, q3 = {x6, x7}, q4 = q0 + 1.f, q5 = q1 + 1.f, q6 = q2 + 1.f, q7 = q3 + 1.f, qa = {a, a};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP8("v_fma_f32") : R8 : "v"(a));
        else if (MODE == 1) asm volatile(REP8_2("v_mul_f32") : R8 : "v"(a));
        else if (MODE == 2) asm volatile(REP8_2("v_add_f32") : R8 : "v"(a));
        else if (MODE == 3) asm volatile(REP8_1("v_exp_f32") : R8 : "v"(a));
        else if (MODE == 4) asm volatile(REP8_1("v_rcp_f32") : R8 : "v"(a));
        else if (MODE == 5)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qa));
        else if (MODE == 6)
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qa));
        else if (MODE == 7)  // DPP add, dependent chain per register
            asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" : R8 : "v"(a));
        else if (MODE == 8)  // fma with an SGPR operand (the blend kernels' splat record lives in SGPRs)
            asm volatile("v_fma_f32 %0, %0, s20, %0\n v_fma_f32 %1, %1, s20, %1\n v_fma_f32 %2, %2, s20, %2\n v_fma_f32 %3, %3, s20, %3\n"
                         "v_fma_f32 %4, %4, s20, %4\n v_fma_f32 %5, %5, s20, %5\n v_fma_f32 %6, %6, s20, %6\n v_fma_f32 %7, %7, s20, %7\n" : R8 : "v"(a) : "s20");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
}
template <int MODE> void run(const char* name)
{
    float* d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000, blocks = 2048;
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 10);
    (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double per_simd = (double)blocks * 4 * iters * 8 / 1024.0;  // wave-instructions per SIMD
    printf("%-34s %8.3f ms -> %.2f ns per wave64 instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
    (void)hipFree(d);
}
int main()
{
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<2>("v_add_f32"); run<3>("v_exp_f32"); run<4>("v_rcp_f32");
    run<5>("v_pk_fma_f32 (2 FMAs each)"); run<6>("v_pk_mul_f32 (2 muls each)"); run<7>("v_add_f32 DPP quad_perm");
    run<8>("v_fma_f32 with SGPR operand");
}
