// Row f-7 (producers of the rasterizer's `rotations` argument on the human model's forward path, every training step):
//   rotation_6d_to_matrix(d6)     /root/reference/hugs/utils/rotations.py:552-573   (hugs_trimlp.py:418: Gram-Schmidt of the MLP's 6-D output)
//   matrix_to_quaternion(matrix)  /root/reference/hugs/utils/rotations.py:94-156    (hugs_trimlp.py:419,518: canonical and LBS-deformed rotations)
// The reference composes them from ~15 and ~40 torch kernels (forward; twice that backward); matrix_to_quaternion also indexes
// with boolean masks twice (x[positive_mask], candidates[one_hot > 0.5]) -- each a device-to-host synchronisation in the middle
// of the step.  Here: one thread per rotation, one kernel per direction, no synchronisation.
//
// matrix_to_quaternion as the reference states it: t = (1+m00+m11+m22, 1+m00-m11-m22, 1-m00+m11-m22, 1-m00-m11+m22),
// q_abs_i = sqrt(t_i) where t_i > 0 else 0 (zero subgradient there), b = argmax q_abs (first maximum), candidate row b of
//   [[q_abs_0^2, m21-m12, m02-m20, m10-m01], [m21-m12, q_abs_1^2, m10+m01, m02+m20],
//    [m02-m20, m10+m01, q_abs_2^2, m12+m21], [m10-m01, m20+m02, m21+m12, q_abs_3^2]]   divided by 2 max(q_abs_b, 0.1).
// Its autograd backward touches only the selected row: out_j = N_j / (2 D), D = max(q_abs_b, 0.1);
//   dL/dN_j = g_j / (2 D),   dL/dD = -sum_j g_j N_j / (2 D^2),   dN_b/dt_b = 1 (t_b > 0),   dD/dt_b = 1 / (2 q_abs_b) (q_abs_b > 0.1).
#include "hgs_common.h"

namespace {

__device__ __forceinline__ int select_row(const float* m, float* t, float* qabs)
{
    t[0] = 1.0f + m[0] + m[4] + m[8], t[1] = 1.0f + m[0] - m[4] - m[8];
    t[2] = 1.0f - m[0] + m[4] - m[8], t[3] = 1.0f - m[0] - m[4] + m[8];
    int b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) qabs[k] = t[k] > 0.0f ? sqrtf(t[k]) : 0.0f;
#pragma unroll
    for (int k = 1; k < 4; ++k) b = qabs[k] > qabs[b] ? k : b;  // first maximum, as torch.argmax
    return b;
}

// the candidate row's three off-diagonal entries: N_j = m[P[b][j][0]] + S[b][j] * m[P[b][j][1]]  (flat 3x3 indices)
__device__ __forceinline__ void row_terms(int b, int j, int& p, int& q, float& s)
{
    // symmetric table: entry (b, j) == entry (j, b)
    const int lo = b < j ? b : j, hi = b < j ? j : b;
    if (lo == 0) {  // (0,1): m21 - m12   (0,2): m02 - m20   (0,3): m10 - m01
        p = hi == 1 ? 7 : hi == 2 ? 2 : 3, q = hi == 1 ? 5 : hi == 2 ? 6 : 1, s = -1.0f;
    } else if (lo == 1) {  // (1,2): m10 + m01   (1,3): m02 + m20
        p = hi == 2 ? 3 : 2, q = hi == 2 ? 1 : 6, s = 1.0f;
    } else {  // (2,3): m12 + m21
        p = 5, q = 7, s = 1.0f;
    }
}

__global__ void __launch_bounds__(256) matrix_to_quaternion_kernel(int n, const float* __restrict__ matrix, float* __restrict__ quat)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m[9], t[4], qabs[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = matrix[9 * (size_t)i + k];
    const int b = select_row(m, t, qabs);
    const float denom = 2.0f * fmaxf(qabs[b], 0.1f);
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p, q;
        float s;
        row_terms(b, j == b ? (b + 1) & 3 : j, p, q, s);
        float N = m[p] + s * m[q];
        if (j == b) N = qabs[b] * qabs[b];
        out[j] = N / denom;
    }
    *reinterpret_cast<float4*>(quat + 4 * (size_t)i) = make_float4(out[0], out[1], out[2], out[3]);
}

__global__ void __launch_bounds__(256)
matrix_to_quaternion_backward_kernel(int n, const float* __restrict__ matrix, const float* __restrict__ dL_dquat, float* __restrict__ dL_dmatrix)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m[9], t[4], qabs[4], d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = matrix[9 * (size_t)i + k], d[k] = 0.0f;
    const int b = select_row(m, t, qabs);
    const float4 g4 = *reinterpret_cast<const float4*>(dL_dquat + 4 * (size_t)i);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float D = fmaxf(qabs[b], 0.1f), inv2D = 1.0f / (2.0f * D);
    float gD = 0.0f, gt = 0.0f;  // dL/dD, dL/dt_b
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p, q;
        float s;
        row_terms(b, j == b ? (b + 1) & 3 : j, p, q, s);
        const float gN = g[j] * inv2D;
        float N;
        if (j == b) {
            N = qabs[b] * qabs[b];
            gt += t[b] > 0.0f ? gN : 0.0f;  // d(sqrt(t)^2)/dt = 1 where t > 0, else the zero subgradient
        } else {
            N = m[p] + s * m[q];
            d[p] += gN, d[q] += s * gN;
        }
        gD -= g[j] * N * inv2D / D;
    }
    if (qabs[b] > 0.1f) gt += gD / (2.0f * qabs[b]);
    // t_b = 1 + s0 m00 + s1 m11 + s2 m22
    d[0] += (b == 0 || b == 1) ? gt : -gt;
    d[4] += (b == 0 || b == 2) ? gt : -gt;
    d[8] += (b == 0 || b == 3) ? gt : -gt;
#pragma unroll
    for (int k = 0; k < 9; ++k) dL_dmatrix[9 * (size_t)i + k] = d[k];
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// b1 = normalize(a1), u = a2 - (b1.a2) b1, b2 = normalize(u), b3 = b1 x b2 (rows of the matrix); normalize = x / max(|x|, 1e-12)
__global__ void __launch_bounds__(256) rotation_6d_to_matrix_kernel(int n, const float* __restrict__ d6, float* __restrict__ matrix)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* a = d6 + 6 * (size_t)i;
    const V3 a1{a[0], a[1], a[2]}, a2{a[3], a[4], a[5]};
    const float n1 = fmaxf(sqrtf(dot(a1, a1)), 1e-12f);
    const V3 b1{a1.x / n1, a1.y / n1, a1.z / n1};  // (divisions, as F.normalize: not a reciprocal and a product)
    const V3 u = a2 - dot(b1, a2) * b1;
    const float nu = fmaxf(sqrtf(dot(u, u)), 1e-12f);
    const V3 b2{u.x / nu, u.y / nu, u.z / nu};
    const V3 b3 = cross(b1, b2);
    float* o = matrix + 9 * (size_t)i;
    o[0] = b1.x, o[1] = b1.y, o[2] = b1.z, o[3] = b2.x, o[4] = b2.y, o[5] = b2.z, o[6] = b3.x, o[7] = b3.y, o[8] = b3.z;
}

__device__ __forceinline__ V3 normalize_backward(V3 y, float norm, V3 g)
{
    // y = x / max(|x|, eps), norm = |x|: (g - y (y.g)) / |x| above eps, g / eps below (the clamp is then a constant)
    return norm > 1e-12f ? (1.0f / norm) * (g - dot(y, g) * y) : 1e12f * g;
}

__global__ void __launch_bounds__(256)
rotation_6d_to_matrix_backward_kernel(int n, const float* __restrict__ d6, const float* __restrict__ dL_dmatrix, float* __restrict__ dL_dd6)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* a = d6 + 6 * (size_t)i;
    const float* G = dL_dmatrix + 9 * (size_t)i;
    const V3 a1{a[0], a[1], a[2]}, a2{a[3], a[4], a[5]}, g1{G[0], G[1], G[2]}, g2{G[3], G[4], G[5]}, g3{G[6], G[7], G[8]};
    const float n1 = sqrtf(dot(a1, a1));
    const V3 b1 = (1.0f / fmaxf(n1, 1e-12f)) * a1;
    const float s = dot(b1, a2);
    const V3 u = a2 - s * b1;
    const float nu = sqrtf(dot(u, u));
    const V3 b2 = (1.0f / fmaxf(nu, 1e-12f)) * u;
    // b3 = b1 x b2
    V3 gb1 = g1 + cross(b2, g3);
    const V3 gb2 = g2 + cross(g3, b1);
    const V3 gu = normalize_backward(b2, nu, gb2);
    // u = a2 - (b1.a2) b1
    const float gub1 = dot(gu, b1);
    const V3 ga2 = gu - gub1 * b1;
    gb1 = gb1 - s * gu - gub1 * a2;
    const V3 ga1 = normalize_backward(b1, n1, gb1);
    float* o = dL_dd6 + 6 * (size_t)i;
    o[0] = ga1.x, o[1] = ga1.y, o[2] = ga1.z, o[3] = ga2.x, o[4] = ga2.y, o[5] = ga2.z;
}

int fail_rot(const char* what)
{
    hgs::set_last_error(what);
    return HGS_ERR_INVALID_ARGUMENT;
}

template <typename... A>
int launch(const char* what, void (*kernel)(A...), int n, void* stream, A... a)
{
    hipLaunchKernelGGL(kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a...);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error(what);
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

}  // namespace

extern "C" int32_t hgs_matrix_to_quaternion(int32_t n, const float* matrix, float* quat, void* stream)
{
    if (n < 0) return fail_rot("matrix_to_quaternion: n < 0");
    if (n == 0) return HGS_OK;
    if (!matrix || !quat || ((uintptr_t)quat & 15) != 0) return fail_rot("matrix_to_quaternion: null pointer, or quat not 16-byte aligned");
    return launch("matrix_to_quaternion: kernel launch failed", matrix_to_quaternion_kernel, n, stream, (int)n, matrix, quat);
}

extern "C" int32_t hgs_matrix_to_quaternion_backward(int32_t n, const float* matrix, const float* dL_dquat, float* dL_dmatrix, void* stream)
{
    if (n < 0) return fail_rot("matrix_to_quaternion_backward: n < 0");
    if (n == 0) return HGS_OK;
    if (!matrix || !dL_dquat || !dL_dmatrix || ((uintptr_t)dL_dquat & 15) != 0)
        return fail_rot("matrix_to_quaternion_backward: null pointer, or dL_dquat not 16-byte aligned");
    return launch("matrix_to_quaternion_backward: kernel launch failed", matrix_to_quaternion_backward_kernel, n, stream, (int)n, matrix,
                  dL_dquat, dL_dmatrix);
}

extern "C" int32_t hgs_rotation_6d_to_matrix(int32_t n, const float* d6, float* matrix, void* stream)
{
    if (n < 0) return fail_rot("rotation_6d_to_matrix: n < 0");
    if (n == 0) return HGS_OK;
    if (!d6 || !matrix) return fail_rot("rotation_6d_to_matrix: null pointer");
    return launch("rotation_6d_to_matrix: kernel launch failed", rotation_6d_to_matrix_kernel, n, stream, (int)n, d6, matrix);
}

extern "C" int32_t hgs_rotation_6d_to_matrix_backward(int32_t n, const float* d6, const float* dL_dmatrix, float* dL_dd6, void* stream)
{
    if (n < 0) return fail_rot("rotation_6d_to_matrix_backward: n < 0");
    if (n == 0) return HGS_OK;
    if (!d6 || !dL_dmatrix || !dL_dd6) return fail_rot("rotation_6d_to_matrix_backward: null pointer");
    return launch("rotation_6d_to_matrix_backward: kernel launch failed", rotation_6d_to_matrix_backward_kernel, n, stream, (int)n, d6,
                  dL_dmatrix, dL_dd6);
}
