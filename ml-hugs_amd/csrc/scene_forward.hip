// Row f-6 (the producer on the input side of the rasterizer): SceneGS.forward, /root/reference/hugs/models/scene.py:147-160 --
//   scales = exp(_scaling)                      (:148, scaling_activation = torch.exp, :42)
//   rotq   = normalize(_rotation)               (:149, torch.nn.functional.normalize: x / max(|x|, 1e-12), :50)
//   opacity = sigmoid(_opacity)                 (:151, :47)
//   shs    = cat(_features_dc, _features_rest)  (:152 -> get_features, :132-138)
// run on every training step right before the render.  The reference spends 5 forward and ~17 backward torch kernels on it
// (0.12 ms of GPU time and 0.26 ms of launches at 200 000 Gaussians); here it is one forward and one backward kernel, both
// plain streaming: 8 floats in / 8 out per Gaussian plus the SH row, which moves as float4 stores (forward) / float4 loads
// (backward) with the [P,1,3] + [P,M-1,3] halves addressed per element.
#include <algorithm>

#include "hgs_common.h"

namespace {

__device__ __forceinline__ float sh_source(const float* __restrict__ dc, const float* __restrict__ rest, size_t p, int e, int row)
{
    return e < 3 ? dc[3 * p + e] : rest[(size_t)(row - 3) * p + (e - 3)];
}

// VEC: the SH row (3M floats) is a whole number of float4s and `shs` is 16-byte aligned
template <bool VEC>
__global__ void __launch_bounds__(256)
scene_forward_kernel(int P, int M, const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ opacity,
                     const float* __restrict__ dc, const float* __restrict__ rest, float* __restrict__ scales, float* __restrict__ rotq,
                     float* __restrict__ opac, float* __restrict__ shs)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int row = 3 * M;
    if (t < (size_t)P) {
#pragma unroll
        for (int k = 0; k < 3; ++k) scales[3 * t + k] = expf(scaling[3 * t + k]);
        const float4 q = *reinterpret_cast<const float4*>(rotation + 4 * t);
        const float n = fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
        *reinterpret_cast<float4*>(rotq + 4 * t) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
        opac[t] = 1.0f / (1.0f + expf(-opacity[t]));
    }
    if (VEC) {
        const int per_row = row / 4;
        if (t < (size_t)P * per_row) {
            const size_t p = t / per_row;
            const int e = (int)(t - p * per_row) * 4;
            reinterpret_cast<float4*>(shs)[t] = make_float4(sh_source(dc, rest, p, e, row), sh_source(dc, rest, p, e + 1, row),
                                                             sh_source(dc, rest, p, e + 2, row), sh_source(dc, rest, p, e + 3, row));
        }
    } else {
        for (size_t i = t; i < (size_t)P * row; i += (size_t)gridDim.x * 256) {
            const size_t p = i / row;
            shs[i] = sh_source(dc, rest, p, (int)(i - p * row), row);
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256)
scene_backward_kernel(int P, int M, const float* __restrict__ rotation, const float* __restrict__ scales, const float* __restrict__ opac,
                      const float* __restrict__ g_scales, const float* __restrict__ g_rotq, const float* __restrict__ g_opac,
                      const float* __restrict__ g_shs, float* __restrict__ d_scaling, float* __restrict__ d_rotation,
                      float* __restrict__ d_opacity, float* __restrict__ d_dc, float* __restrict__ d_rest)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int row = 3 * M;
    if (t < (size_t)P) {
        if (g_scales) {
#pragma unroll
            for (int k = 0; k < 3; ++k) d_scaling[3 * t + k] = g_scales[3 * t + k] * scales[3 * t + k];   // d exp = exp
        }
        if (g_rotq) {
            // y = x / n, n = max(|x|, eps):  dx = (g - y (y.g)) / n above eps, g / eps below it (n is then a constant)
            const float4 x = *reinterpret_cast<const float4*>(rotation + 4 * t), g = *reinterpret_cast<const float4*>(g_rotq + 4 * t);
            const float norm = sqrtf(((x.x * x.x + x.y * x.y) + x.z * x.z) + x.w * x.w);
            float4 d;
            if (norm > 1e-12f) {
                const float inv = 1.0f / norm;
                const float4 y = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
                const float yg = ((y.x * g.x + y.y * g.y) + y.z * g.z) + y.w * g.w;
                d = make_float4((g.x - y.x * yg) * inv, (g.y - y.y * yg) * inv, (g.z - y.z * yg) * inv, (g.w - y.w * yg) * inv);
            } else {
                d = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
            }
            *reinterpret_cast<float4*>(d_rotation + 4 * t) = d;
        }
        if (g_opac) {
            const float o = opac[t];
            d_opacity[t] = g_opac[t] * o * (1.0f - o);
        }
    }
    if (!g_shs) return;
    auto sink = [&](size_t p, int e, float v) {
        if (e < 3) d_dc[3 * p + e] = v;
        else d_rest[(size_t)(row - 3) * p + (e - 3)] = v;
    };
    if (VEC) {
        const int per_row = row / 4;
        if (t < (size_t)P * per_row) {
            const size_t p = t / per_row;
            const int e = (int)(t - p * per_row) * 4;
            const float4 g = reinterpret_cast<const float4*>(g_shs)[t];
            sink(p, e, g.x), sink(p, e + 1, g.y), sink(p, e + 2, g.z), sink(p, e + 3, g.w);
        }
    } else {
        for (size_t i = t; i < (size_t)P * row; i += (size_t)gridDim.x * 256) {
            const size_t p = i / row;
            sink(p, (int)(i - p * row), g_shs[i]);
        }
    }
}

int fail_scene(const char* what)
{
    hgs::set_last_error(what);
    return HGS_ERR_INVALID_ARGUMENT;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

unsigned blocks_for(int P, int M, bool vec)
{
    const size_t threads = vec ? (size_t)P * (size_t)std::max(3 * M / 4, 1) : (size_t)P * (size_t)std::min(3 * M, 16);
    return (unsigned)((std::max(threads, (size_t)P) + 255) / 256);
}

}  // namespace

extern "C" int32_t hgs_scene_forward(int32_t P, int32_t M, const float* scaling, const float* rotation, const float* opacity,
                                     const float* features_dc, const float* features_rest, float* scales, float* rotq,
                                     float* opacities, float* shs, void* stream)
{
    if (P < 0 || M < 1 || M > 64) return fail_scene("scene_forward: need P >= 0 and 1 <= M <= 64");
    if (P == 0) return HGS_OK;
    if (!scaling || !rotation || !opacity || !features_dc || (M > 1 && !features_rest) || !scales || !rotq || !opacities || !shs)
        return fail_scene("scene_forward: null pointer");
    if (!aligned16(rotation) || !aligned16(rotq)) return fail_scene("scene_forward: rotation / rotq must be 16-byte aligned");
    const bool vec = (3 * M) % 4 == 0 && aligned16(shs);
    const dim3 grid(blocks_for(P, M, vec));
    if (vec) hipLaunchKernelGGL(scene_forward_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, P, M, scaling, rotation, opacity, features_dc, features_rest, scales, rotq, opacities, shs);
    else hipLaunchKernelGGL(scene_forward_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, P, M, scaling, rotation, opacity, features_dc, features_rest, scales, rotq, opacities, shs);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("scene_forward: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_scene_backward(int32_t P, int32_t M, const float* rotation, const float* scales, const float* opacities,
                                      const float* dL_dscales, const float* dL_drotq, const float* dL_dopacities, const float* dL_dshs,
                                      float* dL_dscaling, float* dL_drotation, float* dL_dopacity, float* dL_dfeatures_dc,
                                      float* dL_dfeatures_rest, void* stream)
{
    if (P < 0 || M < 1 || M > 64) return fail_scene("scene_backward: need P >= 0 and 1 <= M <= 64");
    if (P == 0) return HGS_OK;
    if ((dL_dscales && (!scales || !dL_dscaling)) || (dL_drotq && (!rotation || !dL_drotation)) || (dL_dopacities && (!opacities || !dL_dopacity)) ||
        (dL_dshs && (!dL_dfeatures_dc || (M > 1 && !dL_dfeatures_rest))))
        return fail_scene("scene_backward: a gradient was given without the tensors it needs");
    if (dL_drotq && (!aligned16(rotation) || !aligned16(dL_drotq) || !aligned16(dL_drotation)))
        return fail_scene("scene_backward: rotation and its gradients must be 16-byte aligned");
    const bool vec = (3 * M) % 4 == 0 && aligned16(dL_dshs);
    const dim3 grid(blocks_for(P, M, vec));
    if (vec) hipLaunchKernelGGL(scene_backward_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, P, M, rotation, scales, opacities, dL_dscales, dL_drotq, dL_dopacities, dL_dshs, dL_dscaling, dL_drotation, dL_dopacity, dL_dfeatures_dc, dL_dfeatures_rest);
    else hipLaunchKernelGGL(scene_backward_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, P, M, rotation, scales, opacities, dL_dscales, dL_drotq, dL_dopacities, dL_dshs, dL_dscaling, dL_drotation, dL_dopacity, dL_dfeatures_dc, dL_dfeatures_rest);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("scene_backward: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}
