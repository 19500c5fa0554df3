// K nearest template vertices of every Gaussian, and the SMPL "ground-truth" LBS weights built
// from them -- the two things the reference gets from pytorch3d's CUDA `knn_points` on every training step:
//   knn_points(points, template_points, K)                                   hugs/models/hugs_wo_trimlp.py:60,99
//   smpl_lbsweight_top_k(lbs_weights, points, template_points, K=6)          hugs/models/hugs_wo_trimlp.py:88-119,
//                                                                            called at hugs_trimlp.py:318,480
// pytorch3d is not in /root/reference (pip dependency); its published contract is restated: squared L2 distances,
// the K smallest per query in ascending order, int64 indices.  Ties go to the lower template index (oracle/knn_oracle.py).
//
// Shape of the problem: n ~ 1e5 queries x m = 6 890 template vertices, K = 6: brute force.  A workgroup owns 64 queries
// (one per lane) and its four waves each scan a quarter of the template -- n/64 waves would leave most SIMDs with one or
// two waves and nothing to hide latency with.  The template vertex of an iteration is the same for every lane of a
// wave, so it is fetched through the scalar cache (s_load_dwordx4, 4 vertices per 3 loads) and used as the scalar
// operand of the VALU math.  A lane keeps its K best (distance, index) pairs sorted in registers; the insertion code
// runs under the exec mask of the few lanes that found a closer vertex.  The four partial lists meet in LDS and wave 0
// merges them in segment order, which preserves the tie rule.
#include <cstdio>

#include "hgs_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* const_f4p;
typedef const __attribute__((address_space(4))) float* const_f32p;

template <int K>
struct Best {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) d[k] = __builtin_inff(), i[k] = -1;
    }
    // strict <: among equal distances the earlier (lower) template index stays in front
    __device__ __forceinline__ void offer(float dist, int idx)
    {
        if (dist < d[K - 1]) {
            d[K - 1] = dist, i[K - 1] = idx;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
                const bool up = d[k] < d[k - 1];
                const float dl = up ? d[k] : d[k - 1], dh = up ? d[k - 1] : d[k];
                const int il = up ? i[k] : i[k - 1], ih = up ? i[k - 1] : i[k];
                d[k - 1] = dl, d[k] = dh, i[k - 1] = il, i[k] = ih;
            }
        }
    }
};

// same operation order as pytorch3d's per-dimension accumulation: ((dx^2 + dy^2) + dz^2); built with -ffp-contract=off
__device__ __forceinline__ float sqdist(float px, float py, float pz, float tx, float ty, float tz)
{
    const float dx = px - tx, dy = py - ty, dz = pz - tz;
    return (dx * dx + dy * dy) + dz * dz;
}

constexpr int KNN_SPLIT = 4;  // waves per workgroup = template segments

// scans template vertices [j0, j1) in ascending order; vertex `skip` (or -1) is never offered (a cloud searched against
// itself: a point is not its own neighbour).  `templ` needs only float alignment: vertices up to the first one that
// starts on a 16-byte boundary (vertex j does when j == phase mod 4, phase = float offset of templ inside its 16 bytes)
// are taken one at a time, as is the tail.
template <int K>
__device__ __forceinline__ void scan_template(float px, float py, float pz, const float* __restrict__ templ, int j0, int j1,
                                              int skip, Best<K>& best)
{
    best.init();
    const int phase = (int)(((uintptr_t)templ >> 2) & 3u);
    const int ja = min(j0 + ((phase - j0) & 3), j1);
    for (int j = j0; j < ja; ++j) {
        const_f32p q = (const_f32p)(templ + 3 * (size_t)j);
        if (j != skip) best.offer(sqdist(px, py, pz, q[0], q[1], q[2]), j);
    }
    j0 = ja;
    const int j4 = j0 + ((j1 - j0) & ~3);
    // 4 vertices = 12 floats = three aligned 16-byte scalar loads; the next trip's loads are issued before this trip's
    // arithmetic (two register sets), so the scalar-cache latency runs under it
    v4f a, b, c;
    if (j0 < j4) {
        const_f4p q = (const_f4p)(templ + 3 * (size_t)j0);
        a = q[0], b = q[1], c = q[2];
    }
    for (int j = j0; j < j4; j += 4) {
        const int jn = j + 4 < j4 ? j + 4 : j;  // last trip: reload the same vertices (harmless)
        const_f4p qn = (const_f4p)(templ + 3 * (size_t)jn);
        const v4f an = qn[0], bn = qn[1], cn = qn[2];
        // four independent distance chains and one wave-level test for "nobody improves", the usual case
        float d0 = sqdist(px, py, pz, a.x, a.y, a.z), d1 = sqdist(px, py, pz, a.w, b.x, b.y);
        float d2 = sqdist(px, py, pz, b.z, b.w, c.x), d3 = sqdist(px, py, pz, c.y, c.z, c.w);
        a = an, b = bn, c = cn;
        const unsigned ds = (unsigned)(skip - j);
        if (ds < 4u) {  // at most one trip per lane
            const float inf = __builtin_inff();
            d0 = ds == 0u ? inf : d0, d1 = ds == 1u ? inf : d1, d2 = ds == 2u ? inf : d2, d3 = ds == 3u ? inf : d3;
        }
        const float dmin = fminf(fminf(d0, d1), fminf(d2, d3));
        if (__builtin_amdgcn_ballot_w64(dmin < best.d[K - 1]) == 0ull) continue;
        best.offer(d0, j);
        best.offer(d1, j + 1);
        best.offer(d2, j + 2);
        best.offer(d3, j + 3);
    }
    for (int j = j4; j < j1; ++j) {
        const_f32p q = (const_f32p)(templ + 3 * (size_t)j);
        if (j != skip) best.offer(sqdist(px, py, pz, q[0], q[1], q[2]), j);
    }
}

// The workgroup's search: returns (in wave 0 only, `true`) the K nearest template vertices of point blockIdx.x*64+lane.
template <int K, bool SKIP_SELF = false>
__device__ __forceinline__ bool workgroup_knn(int n, const float* __restrict__ points, int m, const float* __restrict__ templ,
                                              Best<K>& best, int& point)
{
    __shared__ float sh_d[KNN_SPLIT - 1][K][64];
    __shared__ int sh_i[KNN_SPLIT - 1][K][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    point = blockIdx.x * 64 + lane;
    const int pc = point < n ? point : n - 1;  // every lane scans (wave-uniform loads); only valid lanes store
    const int seg = (((m + KNN_SPLIT - 1) / KNN_SPLIT) + 3) & ~3;
    const int j0 = min(w * seg, m), j1 = min(j0 + seg, m);
    scan_template<K>(points[3 * (size_t)pc], points[3 * (size_t)pc + 1], points[3 * (size_t)pc + 2], templ, j0, j1,
                     SKIP_SELF ? pc : -1, best);
    if (w > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh_d[w - 1][k][lane] = best.d[k], sh_i[w - 1][k][lane] = best.i[k];
    }
    __syncthreads();
    if (w > 0) return false;
    // segments in index order, each list ascending: an equal distance from a later segment stays behind (tie rule)
#pragma unroll
    for (int q = 0; q < KNN_SPLIT - 1; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k) best.offer(sh_d[q][k][lane], sh_i[q][k][lane]);
    return true;
}

template <int K>
__global__ void __launch_bounds__(64 * KNN_SPLIT)
knn_kernel(int n, const float* __restrict__ points, int m, const float* __restrict__ templ, float* __restrict__ dists,
           int64_t* __restrict__ idx)
{
    Best<K> best;
    int i;
    if (!workgroup_knn<K>(n, points, m, templ, best, i) || i >= n) return;
#pragma unroll
    for (int k = 0; k < K; ++k) dists[(size_t)i * K + k] = best.d[k], idx[(size_t)i * K + k] = (int64_t)best.i[k];
}

// smpl_lbsweight_top_k fused behind the search (hugs_wo_trimlp.py:101-119):
//   conf_k = [exp(-sum_j |w[idx_k][j] - w[idx_0][j]| / (2 * 0.1^2)) > 0.9]
//   wgt_k  = exp(-dist_k) * conf_k;  wgt_k /= sum_k wgt_k
//   out_weights[j] = sum_k wgt_k * w[idx_k][j];  out_dist = sum_k wgt_k * dist_k
template <int K>
__global__ void __launch_bounds__(64 * KNN_SPLIT)
lbsweight_top_k_kernel(int n, const float* __restrict__ points, int m, const float* __restrict__ templ,
                       const float* __restrict__ lbs_weights, int J, float* __restrict__ out_dist,
                       float* __restrict__ out_weights)
{
    Best<K> best;
    int i;
    if (!workgroup_knn<K>(n, points, m, templ, best, i) || i >= n) return;
    const float weight_std2 = (float)(2.0 * 0.1 * 0.1);
    const float* w0 = lbs_weights + (size_t)best.i[0] * J;
    float wgt[K], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float* wk = lbs_weights + (size_t)best.i[k] * J;
        float l1 = 0.0f;
#pragma unroll 8
        for (int j = 0; j < J; ++j) l1 += fabsf(wk[j] - w0[j]);  // unrolled: eight rows' loads in flight per trip
        const float conf = expf(-l1 / weight_std2) > 0.9f ? 1.0f : 0.0f;
        wgt[k] = expf(-best.d[k]) * conf;
        sum += wgt[k];
    }
    float dist = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wgt[k] = wgt[k] / sum;
        dist += wgt[k] * best.d[k];
    }
    out_dist[i] = dist;
#pragma unroll 4
    for (int j = 0; j < J; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += wgt[k] * lbs_weights[(size_t)best.i[k] * J + j];
        out_weights[(size_t)i * J + j] = acc;
    }
}

// simple_knn's distCUDA2 -- mean squared distance of every point of a cloud to its three nearest
// OTHER points (scene.py:181, initial scales).  Upstream sorts by Morton code and searches boxes; at initialisation
// sizes (1e5 points) the brute-force scan above does the n^2 distances in a few milliseconds, exactly.
__global__ void __launch_bounds__(64 * KNN_SPLIT)
mean_dist3_kernel(int n, const float* __restrict__ points, float* __restrict__ mean_dist2)
{
    Best<3> best;
    int i;
    if (!workgroup_knn<3, true>(n, points, n, points, best, i) || i >= n) return;
    mean_dist2[i] = ((best.d[0] + best.d[1]) + best.d[2]) / 3.0f;
}

int fail_knn(const char* what)
{
    hgs::set_last_error(what);
    return HGS_ERR_INVALID_ARGUMENT;
}

template <template <int> class Launch, typename... A>
int dispatch_k(int K, A... a)
{
    switch (K) {
        case 1: return Launch<1>::go(a...);
        case 2: return Launch<2>::go(a...);
        case 3: return Launch<3>::go(a...);
        case 4: return Launch<4>::go(a...);
        case 5: return Launch<5>::go(a...);
        case 6: return Launch<6>::go(a...);
        case 7: return Launch<7>::go(a...);
        case 8: return Launch<8>::go(a...);
        default: return fail_knn("K must be between 1 and 8");
    }
}

template <int K>
struct LaunchKnn {
    static int go(int n, const float* p, int m, const float* t, float* d, int64_t* idx, hipStream_t st)
    {
        hipLaunchKernelGGL(knn_kernel<K>, dim3((n + 63) / 64), dim3(64 * KNN_SPLIT), 0, st, n, p, m, t, d, idx);
        return HGS_OK;
    }
};
template <int K>
struct LaunchLbs {
    static int go(int n, const float* p, int m, const float* t, const float* w, int J, float* od, float* ow, hipStream_t st)
    {
        hipLaunchKernelGGL(lbsweight_top_k_kernel<K>, dim3((n + 63) / 64), dim3(64 * KNN_SPLIT), 0, st, n, p, m, t, w, J, od, ow);
        return HGS_OK;
    }
};

}  // namespace

extern "C" int32_t hgs_knn_points(int32_t n, const float* points, int32_t m, const float* template_points, int32_t K,
                                  float* dists, int64_t* idx, void* stream)
{
    if (n < 0 || m < K || K < 1) return fail_knn("knn_points: need n >= 0 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !dists || !idx) return fail_knn("knn_points: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("knn_points: template_points must be float-aligned");
    if (int rc = dispatch_k<LaunchKnn>(K, n, points, m, template_points, dists, idx, (hipStream_t)stream)) return rc;
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("knn_points: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_smpl_lbsweight_top_k(int32_t n, const float* points, int32_t m, const float* template_points,
                                            const float* lbs_weights, int32_t J, int32_t K, float* out_dist,
                                            float* out_weights, void* stream)
{
    if (n < 0 || m < K || K < 1 || J < 1) return fail_knn("smpl_lbsweight_top_k: need n >= 0, J >= 1 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !lbs_weights || !out_dist || !out_weights) return fail_knn("smpl_lbsweight_top_k: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("smpl_lbsweight_top_k: template_points must be float-aligned");
    if (int rc = dispatch_k<LaunchLbs>(K, n, points, m, template_points, lbs_weights, J, out_dist, out_weights, (hipStream_t)stream)) return rc;
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("smpl_lbsweight_top_k: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_dist_cuda2(int32_t n, const float* points, float* mean_dist2, void* stream)
{
    if (n < 0) return fail_knn("distCUDA2: n < 0");
    if (n == 0) return HGS_OK;
    if (n < 4) return fail_knn("distCUDA2: needs at least 4 points (three neighbours besides the point itself)");
    if (!points || !mean_dist2) return fail_knn("distCUDA2: null pointer");
    if (((uintptr_t)points & 3) != 0) return fail_knn("distCUDA2: points must be float-aligned");
    hipLaunchKernelGGL(mean_dist3_kernel, dim3((n + 63) / 64), dim3(64 * KNN_SPLIT), 0, (hipStream_t)stream, n, points, mean_dist2);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("distCUDA2: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}
