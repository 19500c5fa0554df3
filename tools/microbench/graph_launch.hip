// What a HIP graph buys for a frame of ~7 short dependent kernels on gfx950 / ROCm 7: host time per frame and the GPU-side
// distance between consecutive kernels, against plain stream launches.  (Round 5, DESIGN section 4.1: the decision between
// a graph per frame shape and a thinner eager host path.)
//   eager            : 7 hipLaunchKernelGGL per frame
//   graph            : the same 7 launches captured once, one hipGraphLaunch per frame
//   graph+setparams  : hipGraphExecKernelNodeSetParams on every node before each launch (pointers that change per frame)
// Each variant twice: kernels that do nothing (host cost per frame) and kernels that spin ~5 us (GPU-bound: frame period
// minus 7 x kernel time = what the launches' dependencies cost on the device).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { void* p[24]; int v[16]; };   // 256 bytes of arguments, like the rasterizer's launch structs

__global__ void __launch_bounds__(256) spin_kernel(Big b, float* out, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += (float)b.v[0];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* out;
    CK(hipMalloc(&out, 256));
    CK(hipMemset(out, 0, 256));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const int K = 7, FRAMES = 3000;
    for (double kernel_us : {0.0, 5.0, 15.0}) {
        const long long ticks = (long long)(kernel_us * rate_khz / 1000.0);
        Big b{};
        auto launch_frame = [&](hipStream_t s) {
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, b, out, ticks);
        };
        // eager
        for (int i = 0; i < 200; ++i) launch_frame(st);
        CK(hipStreamSynchronize(st));
        double t0 = now_us();
        for (int i = 0; i < FRAMES; ++i) launch_frame(st);
        double t_host = now_us() - t0;
        CK(hipStreamSynchronize(st));
        double t_all = now_us() - t0;
        printf("kernel %4.1f us  eager            : host %6.2f us/frame  period %6.2f us/frame  (7 kernels = %5.1f us)\n", kernel_us, t_host / FRAMES,
               t_all / FRAMES, 7 * kernel_us);
        // graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_frame(st);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        t0 = now_us();
        for (int i = 0; i < FRAMES; ++i) CK(hipGraphLaunch(ge, st));
        t_host = now_us() - t0;
        CK(hipStreamSynchronize(st));
        t_all = now_us() - t0;
        printf("kernel %4.1f us  graph            : host %6.2f us/frame  period %6.2f us/frame\n", kernel_us, t_host / FRAMES, t_all / FRAMES);
        // graph + set params on every node
        size_t n = 0;
        CK(hipGraphGetNodes(g, nullptr, &n));
        std::vector<hipGraphNode_t> nodes(n);
        CK(hipGraphGetNodes(g, nodes.data(), &n));
        std::vector<hipKernelNodeParams> params(n);
        for (size_t k = 0; k < n; ++k) CK(hipGraphKernelNodeGetParams(nodes[k], &params[k]));
        t0 = now_us();
        for (int i = 0; i < FRAMES; ++i) {
            for (size_t k = 0; k < n; ++k) CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &params[k]));
            CK(hipGraphLaunch(ge, st));
        }
        t_host = now_us() - t0;
        CK(hipStreamSynchronize(st));
        t_all = now_us() - t0;
        printf("kernel %4.1f us  graph+setparams  : host %6.2f us/frame  period %6.2f us/frame\n", kernel_us, t_host / FRAMES, t_all / FRAMES);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // a 256-byte pinned -> device copy per frame (a frame descriptor the kernels would read their pointers from)
    void *hp, *dp;
    CK(hipHostMalloc(&hp, 256, 0));
    CK(hipMalloc(&dp, 256));
    double t0 = now_us();
    for (int i = 0; i < 3000; ++i) CK(hipMemcpyAsync(dp, hp, 256, hipMemcpyHostToDevice, st));
    double t_host = now_us() - t0;
    CK(hipStreamSynchronize(st));
    double t_all = now_us() - t0;
    printf("256-byte pinned->device hipMemcpyAsync: host %6.2f us  period %6.2f us\n", t_host / 3000, t_all / 3000);
    return 0;
}
