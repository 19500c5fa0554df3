// Device-to-device copy rate of a few kernel shapes (which one bench.py's peak_measured should use).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/copy_bw.hip -o tools/microbench/bin/copy_bw && tools/microbench/bin/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_k(float4* __restrict__ dst_, const float4* __restrict__ src_, size_t n)
{
    f4* dst = reinterpret_cast<f4*>(dst_);
    const f4* src = reinterpret_cast<const f4*>(src_);
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]);
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
static void run(const char* name, float4* d, const float4* s, size_t bytes, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((copy_k<U, NT>), dim3(blocks), dim3(256), 0, 0, d, s, bytes / 16);
    hipEventRecord(a, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((copy_k<U, NT>), dim3(blocks), dim3(256), 0, 0, d, s, bytes / 16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %6d : %8.1f GB/s\n", name, blocks, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = 1ull << 30;
    float4 *s, *d;
    hipMalloc((void**)&s, bytes), hipMalloc((void**)&d, bytes);
    hipMemset(s, 1, bytes), hipMemset(d, 0, bytes);
    for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 262144}) {
        run<1, false>("u1", d, s, bytes, blocks);
        run<4, false>("u4", d, s, bytes, blocks);
        run<4, true>("u4 nontemporal", d, s, bytes, blocks);
        run<8, true>("u8 nontemporal", d, s, bytes, blocks);
    }
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int k = 0; k < 10; ++k) hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s               : %8.1f GB/s\n", "hipMemcpyAsync D2D", 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
    return 0;
}
