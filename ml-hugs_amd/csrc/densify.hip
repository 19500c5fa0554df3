// the trainer-side consumers of the rasterizer's outputs, fused.
// The reference runs, every training step, four boolean-indexed torch ops
//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                     gs_trainer.py:407-410 / 430-433
//   xyz_gradient_accum[vis] += || viewspace_points.grad[:n][vis, :2] ||      scene.py:460-462 / hugs_trimlp.py:880-882
//   denom[vis] += 1
// (each a gather + scatter with a host-visible nonzero()); here it is one pass over the n Gaussians.
#include <cstdio>

#include "hgs_common.h"

namespace {
__global__ void __launch_bounds__(256)
densification_stats_kernel(int n, const float* __restrict__ grad2d /*[>=n,3]*/, const int32_t* __restrict__ radii,
                           const uint8_t* __restrict__ visible, float* __restrict__ max_radii2D,
                           float* __restrict__ xyz_gradient_accum, float* __restrict__ denom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !visible[i]) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)radii[i]);
    const float gx = grad2d[3 * (size_t)i], gy = grad2d[3 * (size_t)i + 1];
    xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}
}  // namespace

extern "C" int32_t hgs_densification_stats(int32_t n, const float* viewspace_grad, const int32_t* radii,
                                           const uint8_t* visibility_filter, float* max_radii2D,
                                           float* xyz_gradient_accum, float* denom, void* stream)
{
    if (n < 0 || (n > 0 && (!viewspace_grad || !radii || !visibility_filter || !max_radii2D || !xyz_gradient_accum || !denom)))
    {
        hgs::set_last_error("densification_stats: n must be >= 0 and all six arrays non-null");
        return HGS_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return HGS_OK;
    hipLaunchKernelGGL(densification_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, viewspace_grad,
                       radii, visibility_filter, max_radii2D, xyz_gradient_accum, denom);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char msg[256];
        snprintf(msg, sizeof msg, "densification_stats: %s", hipGetErrorString(e));
        hgs::set_last_error(msg);
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}
