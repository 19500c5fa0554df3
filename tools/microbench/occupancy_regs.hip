// Workgroups (256 threads) resident per CU as a function of the registers a kernel allocates (clobbers force the allocation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int V, int S, int KB = 0>
__global__ void __launch_bounds__(256) spin(unsigned long long* t)
{
    __shared__ char lds[KB * 1024 + 4];
    lds[threadIdx.x & 3] = 1;
    __syncthreads();
    if constexpr (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (V == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if constexpr (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (S == 80) asm volatile("s_mov_b32 s70, 0" ::: "s70");
    if constexpr (S == 96) asm volatile("s_mov_b32 s88, 0" ::: "s88");
    if constexpr (S == 106) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    while (wall_clock64() - t0 < 2000ull) {}
}
template <int V, int S, int KB = 0>
void run()
{
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<V, S, KB>, 256, 0);
    const int grid = 256 * 16;
    unsigned long long* d;
    (void)hipMalloc(&d, grid * 8);
    (void)hipMemset(d, 0, grid * 8);
    hipLaunchKernelGGL((spin<V, S, KB>), dim3(grid), dim3(256), 0, 0, d);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto v : h) first += (v - t0 < 500ull);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)spin<V, S, KB>);
    printf("LDS %2d KB, clobber up to v%-3d s%-3d (numRegs %d): occupancy API %d workgroups/CU; measured %.2f per CU\n", KB, V - 1, S, fa.numRegs, nb, first / 256.0);
    (void)hipFree(d);
}
int main()
{
    run<32, 32>(); run<64, 32>(); run<72, 32>(); run<96, 32>(); run<128, 32>();
    run<32, 80>(); run<32, 96>(); run<32, 106>(); run<64, 96>(); run<64, 106>(); run<72, 106>();
    run<72, 106, 16>(); run<64, 80, 16>(); run<72, 32, 16>(); run<32, 106, 16>(); run<72, 106, 8>();
    return 0;
}
