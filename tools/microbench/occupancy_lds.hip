// How many 256-thread workgroups of a kernel with N KB of static LDS does one CU of this GPU hold at once?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor, and measured: every workgroup records the wall clock at its start and spins
// for ~20 us; workgroups that start within the first microsecond were resident together.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int KB>
__global__ void __launch_bounds__(256) spin(unsigned long long* t, int vg)
{
    __shared__ char lds[KB * 1024];
    lds[threadIdx.x] = (char)vg;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    while (wall_clock64() - t0 < 2000ull) {}
    if (lds[(threadIdx.x + 1) & 255] == 77 && vg == 123456) t[0] = 0;
}
template <int KB>
void run(const char* name)
{
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<KB>, 256, 0);
    const int grid = 256 * 16;
    unsigned long long* d;
    hipMalloc(&d, grid * 8);
    hipMemset(d, 0, grid * 8);
    hipLaunchKernelGGL(spin<KB>, dim3(grid), dim3(256), 0, 0, d, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto v : h) first += (v - t0 < 500ull);   // started within 5 us of the first
    printf("%s: %2d KB static LDS: occupancy API %d workgroups/CU; measured %d workgroups resident at once = %.2f per CU\n", name, KB, nb, first, first / 256.0);
    hipFree(d);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d, maxThreadsPerMultiProcessor %d\n", p.name,
           p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
    run<1>("spin"); run<4>("spin"); run<8>("spin"); run<16>("spin"); run<20>("spin"); run<32>("spin"); run<64>("spin");
    return 0;
}
