// SURVEY.md 8f row f-2, second half: the skinning step of the human model's learned LBS, fused.
//   lbs_extra, /root/reference/hugs/models/modules/lbs.py:19-73 (called every training step, hugs_trimlp.py:477-489):
//       T_i = sum_j W[i,j] A_j            (:60-66, a [n,J] x [J,16] matmul)
//       verts_i = (T_i [v_i, 1])[:3]      (:68-73, cat + batched matmul + slice)
//   and the rotation product that consumes T right after it (hugs_trimlp.py:517):  rot_i = T_i[:3,:3] R_i
// The reference spends ~8 small torch kernels on this per step, forward and again in autograd's backward; here it is one
// forward kernel and two backward kernels (+ a 384-thread reduction), all bandwidth-bound on ~200 bytes per Gaussian.
//
// Forward / point-side backward: one thread per Gaussian; the joint transforms are wave-uniform, so they are read through
// the scalar cache and used as scalar operands (no LDS staging).
// dL/dA_j = sum_i W[i,j] G_i is a [J,n] x [n,16] contraction over all Gaussians -- the one dense contraction on this path:
// it runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32), every wave accumulating a [32 joints, 16] tile over
// its share of the points, workgroup partials reduced in a fixed order (no float atomics: deterministic gradients).
#include "hgs_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* const_f32p;

constexpr int LBS_MAX_J = 32;

__global__ void __launch_bounds__(256)
lbs_skin_forward_kernel(int n, int J, const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ v,
                        const float* __restrict__ rotmat, float* __restrict__ T_out, float* __restrict__ verts,
                        float* __restrict__ rot_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const_f32p As = (const_f32p)A;
    float T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = 0.0f;
    const float* w = W + (size_t)i * J;
    for (int j = 0; j < J; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = __builtin_fmaf(wj, As[16 * j + k], T[k]);
    }
    float4* To = reinterpret_cast<float4*>(T_out + 16 * (size_t)i);
#pragma unroll
    for (int r = 0; r < 4; ++r) To[r] = make_float4(T[4 * r], T[4 * r + 1], T[4 * r + 2], T[4 * r + 3]);
    const float x = v[3 * (size_t)i], y = v[3 * (size_t)i + 1], z = v[3 * (size_t)i + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) verts[3 * (size_t)i + r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
    if (rotmat) {
        float R[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rotmat[9 * (size_t)i + k];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                rot_out[9 * (size_t)i + 3 * r + c] = (T[4 * r] * R[c] + T[4 * r + 1] * R[3 + c]) + T[4 * r + 2] * R[6 + c];
    }
}

// Point-side backward: G_i = dL/dT_i (everything that reaches the blended transform), dL/dW[i,:], dL/dv_i, dL/dR_i.
__global__ void __launch_bounds__(256)
lbs_skin_backward_points_kernel(int n, int J, const float* __restrict__ A, const float* __restrict__ v,
                                const float* __restrict__ rotmat, const float* __restrict__ T_in,
                                const float* __restrict__ g_verts, const float* __restrict__ g_T,
                                const float* __restrict__ g_rot, float* __restrict__ G_out, float* __restrict__ dW,
                                float* __restrict__ dv, float* __restrict__ d_rotmat)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const_f32p As = (const_f32p)A;
    float G[16], T[16];
    const float4* Ti = reinterpret_cast<const float4*>(T_in + 16 * (size_t)i);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 t = Ti[r];
        T[4 * r] = t.x, T[4 * r + 1] = t.y, T[4 * r + 2] = t.z, T[4 * r + 3] = t.w;
    }
    if (g_T) {
        const float4* gi = reinterpret_cast<const float4*>(g_T + 16 * (size_t)i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 t = gi[r];
            G[4 * r] = t.x, G[4 * r + 1] = t.y, G[4 * r + 2] = t.z, G[4 * r + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) G[k] = 0.0f;
    }
    float gv[3] = {0.f, 0.f, 0.f};
    if (g_verts) {
        const float vh[4] = {v[3 * (size_t)i], v[3 * (size_t)i + 1], v[3 * (size_t)i + 2], 1.0f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            gv[r] = g_verts[3 * (size_t)i + r];
#pragma unroll
            for (int c = 0; c < 4; ++c) G[4 * r + c] = __builtin_fmaf(gv[r], vh[c], G[4 * r + c]);  // verts = T[:3,:] [v,1]
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dv[3 * (size_t)i + c] = (T[c] * gv[0] + T[4 + c] * gv[1]) + T[8 + c] * gv[2];  // T[:3,:3]^T gv
    if (rotmat) {
        float R[9], gR[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rotmat[9 * (size_t)i + k], gR[k] = g_rot ? g_rot[9 * (size_t)i + k] : 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // rot = T33 R:  dT33 = gR R^T,  dR = T33^T gR
                G[4 * r + c] += (gR[3 * r] * R[3 * c] + gR[3 * r + 1] * R[3 * c + 1]) + gR[3 * r + 2] * R[3 * c + 2];
                d_rotmat[9 * (size_t)i + 3 * r + c] = (T[r] * gR[c] + T[4 + r] * gR[3 + c]) + T[8 + r] * gR[6 + c];
            }
    }
    float4* Go = reinterpret_cast<float4*>(G_out + 16 * (size_t)i);
#pragma unroll
    for (int r = 0; r < 4; ++r) Go[r] = make_float4(G[4 * r], G[4 * r + 1], G[4 * r + 2], G[4 * r + 3]);
    float* dw = dW + (size_t)i * J;
    for (int j = 0; j < J; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_fmaf(G[k], As[16 * j + k], acc);
        dw[j] = acc;
    }
}

// dL/dA partials on the matrix cores.  v_mfma_f32_16x16x4_f32: D[i][j] += sum_k a[i][k] b[k][j], lane l supplying
// a[i = l % 16][k = l / 16] and b[k = l / 16][j = l % 16], and holding D[4 (l / 16) + r][l % 16] in result register r.
// Here i = joint, k = one of 4 Gaussians, j = component of the 4x4 transform: a = W[point][joint], b = G[point][component].
constexpr int LBS_PARTIAL_BLOCKS = 256;
__global__ void __launch_bounds__(256)
lbs_skin_backward_joints_kernel(int n, int J, const float* __restrict__ W, const float* __restrict__ G,
                                float* __restrict__ partial /*[gridDim.x][32][16]*/)
{
    __shared__ float red[4][32][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + w, waves = gridDim.x * 4;
    const int sub = lane >> 4, col = lane & 15;
    v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int base = wave * 4; base < n; base += waves * 4) {
        const int p = base + sub;
        const bool ok = p < n;
        const float b = ok ? G[16 * (size_t)p + col] : 0.0f;
        const float a0 = ok && col < J ? W[(size_t)p * J + col] : 0.0f;
        const float a1 = ok && col + 16 < J ? W[(size_t)p * J + col + 16] : 0.0f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[w][4 * sub + r][col] = acc0[r];
        red[w][16 + 4 * sub + r][col] = acc1[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 16; e += 256) {
        const int j = e >> 4, k = e & 15;
        partial[(size_t)blockIdx.x * 512 + e] = ((red[0][j][k] + red[1][j][k]) + red[2][j][k]) + red[3][j][k];
    }
}

__global__ void __launch_bounds__(512)
lbs_skin_backward_reduce_kernel(int blocks, int J, const float* __restrict__ partial, float* __restrict__ dA)
{
    const int e = threadIdx.x;  // (joint, component)
    if ((e >> 4) >= J) return;
    float acc = 0.0f;
    for (int b = 0; b < blocks; ++b) acc += partial[(size_t)b * 512 + e];
    dA[e] = acc;
}

int fail_lbs(const char* what)
{
    hgs::set_last_error(what);
    return HGS_ERR_INVALID_ARGUMENT;
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return HGS_OK;
    char msg[256];
    snprintf(msg, sizeof msg, "%s: %s", what, hipGetErrorString(e));
    hgs::set_last_error(msg);
    return HGS_ERR_HIP;
}

int lbs_blocks(int n) { return n < 256 * LBS_PARTIAL_BLOCKS ? (n + 255) / 256 : LBS_PARTIAL_BLOCKS; }

}  // namespace

extern "C" int32_t hgs_lbs_skin_forward(int32_t n, int32_t J, const float* A, const float* weights, const float* v,
                                        const float* rotmat, float* T, float* verts, float* rot_out, void* stream)
{
    if (n < 0 || J < 1 || J > LBS_MAX_J) return fail_lbs("lbs_skin: need n >= 0 and 1 <= J <= 32 joints");
    if (n == 0) return HGS_OK;
    if (!A || !weights || !v || !T || !verts || (rotmat && !rot_out)) return fail_lbs("lbs_skin: null pointer");
    if (((uintptr_t)T & 15) != 0) return fail_lbs("lbs_skin: T must be 16-byte aligned");
    hipLaunchKernelGGL(lbs_skin_forward_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, J, A, weights, v, rotmat, T,
                       verts, rot_out);
    return check_launch("lbs_skin forward");
}

extern "C" size_t hgs_lbs_skin_backward_workspace(int32_t n, int32_t J)
{
    (void)J;
    return hgs::align_up(sizeof(float) * 16 * (size_t)(n < 1 ? 1 : n)) + sizeof(float) * 512 * (size_t)lbs_blocks(n < 1 ? 1 : n);
}

extern "C" int32_t hgs_lbs_skin_backward(int32_t n, int32_t J, const float* A, const float* weights, const float* v,
                                         const float* rotmat, const float* T, const float* dL_dverts, const float* dL_dT,
                                         const float* dL_drot, float* dL_dA, float* dL_dweights, float* dL_dv,
                                         float* dL_drotmat, void* workspace, void* stream)
{
    if (n < 0 || J < 1 || J > LBS_MAX_J) return fail_lbs("lbs_skin backward: need n >= 0 and 1 <= J <= 32 joints");
    if (!dL_dA) return fail_lbs("lbs_skin backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(dL_dA, 0, sizeof(float) * 16 * J, st) != hipSuccess) return check_launch("lbs_skin backward");
        return HGS_OK;
    }
    if (!A || !weights || !v || !T || !dL_dweights || !dL_dv || !workspace || (rotmat && !dL_drotmat))
        return fail_lbs("lbs_skin backward: null pointer");
    if ((((uintptr_t)T | (uintptr_t)workspace | (uintptr_t)dL_dT) & 15) != 0) return fail_lbs("lbs_skin backward: T, dL_dT and workspace must be 16-byte aligned");
    float* G = (float*)workspace;
    float* partial = (float*)((char*)workspace + hgs::align_up(sizeof(float) * 16 * (size_t)n));
    const int blocks = lbs_blocks(n);
    hipLaunchKernelGGL(lbs_skin_backward_points_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, J, A, v, rotmat, T, dL_dverts, dL_dT,
                       dL_drot, G, dL_dweights, dL_dv, dL_drotmat);
    hipLaunchKernelGGL(lbs_skin_backward_joints_kernel, dim3(blocks), dim3(256), 0, st, n, J, weights, G, partial);
    hipLaunchKernelGGL(lbs_skin_backward_reduce_kernel, dim3(1), dim3(512), 0, st, blocks, J, partial, dL_dA);
    return check_launch("lbs_skin backward");
}
