#include <hip/hip_runtime.h>
#include <cstdio>
// exact instruction sequences via inline asm; 8 independent registers, 8 instructions per loop body (x UNROLL)
#define R8 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = a;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f q0 = {x0, x1}, q1 = {x2, x3}, q2 = {x4, x5}, q3 = {x6, x7}, qy = {a, a};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)  // VOP2 cndmask, vcc constant
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n" : R8 : "v"(y) : "vcc");
        else if (MODE == 1)  // VOPC e32 -> vcc, 8 of them
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cmp_gt_f32 vcc, %8, %1\n v_cmp_gt_f32 vcc, %8, %2\n v_cmp_gt_f32 vcc, %8, %3\n"
                         "v_cmp_gt_f32 vcc, %8, %4\n v_cmp_gt_f32 vcc, %8, %5\n v_cmp_gt_f32 vcc, %8, %6\n v_cmp_gt_f32 vcc, %8, %7\n" : R8 : "v"(y) : "vcc");
        else if (MODE == 2)  // VOP3 cndmask with an SGPR-pair mask
            asm volatile("v_cmp_gt_f32 s[20:21], %8, %0\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n"
                         "v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %0, s[20:21]\n" : R8 : "v"(y) : "s20", "s21");
        else if (MODE == 3)  // 8 cndmask e32 vcc, vcc set outside the loop
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n" : R8 : "v"(y) : );
        else if (MODE == 4)  // e64 encoding but reading vcc
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n"
                         "v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc\n" : R8 : "v"(y) : "vcc");
        else if (MODE == 5)  // cndmask vcc separated by fmas
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %1, %1, %8, %2\n v_cndmask_b32 %2, %2, %3, vcc\n v_fma_f32 %3, %3, %8, %4\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_fma_f32 %5, %5, %8, %6\n v_cndmask_b32 %6, %6, %7, vcc\n v_fma_f32 %7, %7, %8, %0\n" : R8 : "v"(y) : "vcc");
        else if (MODE == 6)  // independent cndmasks (no register shared between neighbours)
            asm volatile("v_cmp_gt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : R8 : "v"(y) : "vcc");
        else if (MODE == 7)  // same register-sharing pattern as mode 0 but with v_max (is it the RAW/WAR pattern?)
            asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n"
                         "v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n" : R8 : "v"(y));
        else if (MODE == 8)  // v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(qy));
        else if (MODE == 9)  // SALU writes vcc, then cndmask
            asm volatile("v_cmp_gt_f32 s[20:21], %8, %0\n s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                         "s_and_b64 vcc, s[20:21], exec\n v_cndmask_b32 %4, %4, %5, vcc\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n" : R8 : "v"(y) : "vcc", "s20", "s21", "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + q0.x + q1.y + q2.x + q3.y;
}
template <int MODE> void run(const char* name, int per_iter)
{
    float* d; (void)hipMalloc(&d, 256 * 8192 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000, blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 10);
    (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double per_simd = (double)blocks * 4 * iters * per_iter / 1024.0;
    printf("%-28s %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / per_simd);
}
int main() { run<0>("cndmask e32 vcc (+1 cmp)", 9); run<2>("cndmask e64 sgpr (+1 cmp)", 9); run<3>("cndmask e32 vcc, no cmp", 8); run<4>("cndmask e64 vcc (+1 cmp)", 9);
             run<5>("cndmask vcc / fma alternating", 9); run<6>("cndmask vcc independent", 9); run<7>("v_max chained pattern", 8); run<8>("v_pk_fma_f32 (4)", 4); run<9>("salu->vcc, cndmask, 3 fma x2", 9); }
