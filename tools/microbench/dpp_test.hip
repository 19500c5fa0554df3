#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void k(float* out)
{
    int lane = threadIdx.x;
    float v = (float)lane;
    out[0 * 64 + lane] = dpp_mov<0xB1>(v);
    out[1 * 64 + lane] = dpp_mov<0x4E>(v);
    out[2 * 64 + lane] = dpp_mov<0x104>(v);
    out[3 * 64 + lane] = dpp_mov<0x114>(v);
    out[4 * 64 + lane] = dpp_mov<0x128>(v);
    out[5 * 64 + lane] = __shfl_xor(v, 16, 64);
    out[6 * 64 + lane] = __shfl_xor(v, 32, 64);
}
int main()
{
    float* d; hipMalloc(&d, 7 * 64 * 4);
    k<<<1, 64>>>(d);
    float h[7 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* n[] = {"xor1", "xor2", "shl4", "shr4", "ror8", "x16", "x32"};
    for (int r = 0; r < 7; ++r) { printf("%s:", n[r]); for (int i = 0; i < 32; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
}
