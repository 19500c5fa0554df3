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

// smpl_lbsmap_top_k fused behind the search (hugs_wo_trimlp.py:47-85, the ablation model without the triplane): the same
// confidence-gated neighbour weights as above, then
//   out_transform[i] = sum_k wgt_k verts_transform[idx_k]   (4x4, 16 floats)     out_info[i] = sum_k wgt_k info[idx_k]   (C floats)
// The neighbour indices and weights are kept for the backward: the reference differentiates through verts_transform and
// addition_info (the search and the weights are constants there too: no_grad search, lbs_weights only enter a `>` gate).
template <int K>
__global__ void __launch_bounds__(64 * KNN_SPLIT)
lbsmap_top_k_kernel(int n, const float* __restrict__ points, int m, const float* __restrict__ templ,
                    const float* __restrict__ lbs_weights, int J, const float* __restrict__ verts_transform,
                    const float* __restrict__ info, int Cc, float* __restrict__ out_dist, float* __restrict__ out_transform,
                    float* __restrict__ out_info, int32_t* __restrict__ out_idx, float* __restrict__ out_wgt)
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
        for (int j = 0; j < J; ++j) l1 += fabsf(wk[j] - w0[j]);
        const float conf = expf(-l1 / weight_std2) > 0.9f ? 1.0f : 0.0f;
        wgt[k] = expf(-best.d[k]) * conf;
        sum += wgt[k];
    }
    float dist = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wgt[k] = wgt[k] / sum;
        dist += wgt[k] * best.d[k];
        out_idx[(size_t)i * K + k] = best.i[k], out_wgt[(size_t)i * K + k] = wgt[k];
    }
    out_dist[i] = dist;
    float4 acc[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4* T = reinterpret_cast<const float4*>(verts_transform + (size_t)best.i[k] * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 t = T[r];
            acc[r].x += wgt[k] * t.x, acc[r].y += wgt[k] * t.y, acc[r].z += wgt[k] * t.z, acc[r].w += wgt[k] * t.w;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) reinterpret_cast<float4*>(out_transform + (size_t)i * 16)[r] = acc[r];
    if (info)
        for (int c = 0; c < Cc; ++c) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) a += wgt[k] * info[(size_t)best.i[k] * Cc + c];
            out_info[(size_t)i * Cc + c] = a;
        }
}

// its backward: dL/dverts_transform[idx_k] += wgt_k dL/dout_transform[i] (and likewise for info): thread = (point, neighbour)
__global__ void __launch_bounds__(256)
lbsmap_backward_kernel(int n, int K, const int32_t* __restrict__ idx, const float* __restrict__ wgt,
                       const float* __restrict__ g_transform, const float* __restrict__ g_info, int Cc,
                       float* __restrict__ d_verts_transform, float* __restrict__ d_info)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)n * K) return;
    const size_t i = t / K;
    const int v = idx[t];
    const float w = wgt[t];
    if (w == 0.0f) return;  // (a neighbour the confidence gate closed)
    if (g_transform)
        for (int e = 0; e < 16; ++e) atomicAdd(&d_verts_transform[(size_t)v * 16 + e], w * g_transform[i * 16 + e]);
    if (g_info)
        for (int c = 0; c < Cc; ++c) atomicAdd(&d_info[(size_t)v * Cc + c], w * g_info[i * Cc + c]);
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

// ---- distCUDA2 on a uniform grid (large clouds) ----------------------------------------------------------------------
// The brute-force scan is exact and fine for an initialisation-sized cloud (1e5 points: milliseconds) but O(n^2); from
// GRID_MIN_POINTS on, the points are counting-sorted into the cells of a uniform grid over their bounding box (about four
// points per cell) and every point searches the shells of cells around its own, nearest first, until its third-best
// distance is certainly smaller than anything a farther shell can hold: every point outside the shells 0..r lies at
// least r cell sizes away.  The candidates' distances are the same fp32 expression as the brute-force scan's and the mean
// is taken over the same three smallest values in ascending order: the result is bit-identical.  A point that has not
// closed after GRID_MAX_RINGS shells (an outlier far from everything) scans the whole cloud, as the brute force does.
constexpr int GRID_MIN_POINTS = 32768, GRID_MAX_RINGS = 24, GRID_MAX_DIM = 1024;
struct GridParams {
    float minx, miny, minz, cell, inv_cell;
    int gx, gy, gz, cells;
    uint32_t lo[3], hi[3];  // bounding box as order-preserving unsigned keys of the floats (atomicMin / atomicMax)
};
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__global__ void grid_init_kernel(GridParams* g)
{
    for (int k = 0; k < 3; ++k) g->lo[k] = 0xFFFFFFFFu, g->hi[k] = 0u;
}
__global__ void __launch_bounds__(256) grid_bbox_kernel(int n, const float* __restrict__ p, GridParams* g)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < n)
        for (int k = 0; k < 3; ++k) lo[k] = hi[k] = float_key(p[3 * (size_t)i + k]);
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d, 64));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d, 64));
        }
        if ((threadIdx.x & 63) == 0) atomicMin(&g->lo[k], lo[k]), atomicMax(&g->hi[k], hi[k]);
    }
}
// one thread: cell size for ~4 points per cell, at most n cells (the counter array has n + 1 entries) and GRID_MAX_DIM per axis
__global__ void grid_setup_kernel(int n, GridParams* g, uint32_t* cell_count)
{
    const float mn[3] = {key_float(g->lo[0]), key_float(g->lo[1]), key_float(g->lo[2])};
    float ext[3];
    for (int k = 0; k < 3; ++k) ext[k] = fmaxf(key_float(g->hi[k]) - mn[k], 0.0f);
    const float longest = fmaxf(fmaxf(ext[0], ext[1]), fmaxf(ext[2], 1e-30f));
    // volume of the box, a degenerate axis counted as one cell thick
    float cell = longest / 2.0f;
    for (int it = 0; it < 64; ++it) {  // shrink the cell until ~n/4 cells (or the per-axis limit) are reached
        const float c = cell * 0.7937005f;  // 2^(-1/3): halves the cell volume
        long long cells = 1;
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            const long long gk = (long long)floorf(ext[k] / c) + 1;
            ok = ok && gk <= GRID_MAX_DIM;
            cells *= gk;
        }
        if (!ok || cells * 4 > (long long)n) break;
        cell = c;
    }
    g->minx = mn[0], g->miny = mn[1], g->minz = mn[2], g->cell = cell, g->inv_cell = 1.0f / cell;
    g->gx = (int)floorf(ext[0] / cell) + 1, g->gy = (int)floorf(ext[1] / cell) + 1, g->gz = (int)floorf(ext[2] / cell) + 1;
    g->cells = g->gx * g->gy * g->gz;
    (void)cell_count;
}
__device__ __forceinline__ void cell_of(const GridParams& g, float x, float y, float z, int& cx, int& cy, int& cz)
{
    cx = min(max((int)((x - g.minx) * g.inv_cell), 0), g.gx - 1);
    cy = min(max((int)((y - g.miny) * g.inv_cell), 0), g.gy - 1);
    cz = min(max((int)((z - g.minz) * g.inv_cell), 0), g.gz - 1);
}
__global__ void __launch_bounds__(256) grid_count_kernel(int n, const float* __restrict__ p, const GridParams* __restrict__ gp,
                                                         uint32_t* __restrict__ cell_id, uint32_t* __restrict__ slot,
                                                         uint32_t* __restrict__ cell_count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const GridParams g = *gp;
    int cx, cy, cz;
    cell_of(g, p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2], cx, cy, cz);
    const uint32_t c = (uint32_t)((cz * g.gy + cy) * g.gx + cx);
    cell_id[i] = c;
    slot[i] = atomicAdd(&cell_count[c], 1u);
}
// exclusive scan of cell_count[0 .. cells] in place (one workgroup, 8 192 entries per trip)
__global__ void __launch_bounds__(1024) grid_scan_kernel(const GridParams* __restrict__ gp, uint32_t* __restrict__ cell_count)
{
    __shared__ uint32_t wsum[16];
    const int cells = gp->cells + 1, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < cells; base += 8192) {
        uint32_t c[8], mine = 0;
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x * 8 + k;
            c[k] = j < cells ? cell_count[j] : 0u, mine += c[k];
        }
        uint32_t incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < w) before += wsum[k];
            total += wsum[k];
        }
        __syncthreads();
        uint32_t run = carry + before + incl - mine;
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x * 8 + k;
            if (j < cells) cell_count[j] = run;
            run += c[k];
        }
        carry += total;
    }
}
__global__ void __launch_bounds__(256) grid_scatter_kernel(int n, const float* __restrict__ p, const uint32_t* __restrict__ cell_id,
                                                           const uint32_t* __restrict__ slot, const uint32_t* __restrict__ cell_start,
                                                           float4* __restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sorted[cell_start[cell_id[i]] + slot[i]] = make_float4(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2], __int_as_float(i));
}
__global__ void __launch_bounds__(256) grid_search_kernel(int n, const GridParams* __restrict__ gp, const uint32_t* __restrict__ cell_start,
                                                          const float4* __restrict__ sorted, float* __restrict__ mean_dist2)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const GridParams g = *gp;
    const float4 me = sorted[t];  // (threads follow the sorted order: a wave's queries are neighbours in space)
    const int self = __float_as_int(me.w);
    int cx, cy, cz;
    cell_of(g, me.x, me.y, me.z, cx, cy, cz);
    Best<3> best;
    best.init();
    auto scan_cell = [&](int x, int y, int z) {
        const uint32_t c = (uint32_t)((z * g.gy + y) * g.gx + x);
        for (uint32_t j = cell_start[c], e = cell_start[c + 1]; j < e; ++j) {
            const float4 q = sorted[j];
            if (__float_as_int(q.w) != self) best.offer(sqdist(me.x, me.y, me.z, q.x, q.y, q.z), 0);
        }
    };
    bool closed = false;
    const int rmax = max(max(g.gx, g.gy), g.gz);
    for (int r = 0; r <= GRID_MAX_RINGS && !closed; ++r) {
        for (int z = max(cz - r, 0); z <= min(cz + r, g.gz - 1); ++z)
            for (int y = max(cy - r, 0); y <= min(cy + r, g.gy - 1); ++y) {
                const bool face = abs(z - cz) == r || abs(y - cy) == r;  // on the shell whatever x is
                if (face) {
                    for (int x = max(cx - r, 0); x <= min(cx + r, g.gx - 1); ++x) scan_cell(x, y, z);
                } else {
                    if (cx - r >= 0) scan_cell(cx - r, y, z);
                    if (r > 0 && cx + r <= g.gx - 1) scan_cell(cx + r, y, z);
                }
            }
        // everything outside the shells 0..r is at least r cell sizes away (a little less: the cell of a point is an fp32 floor)
        const float reach = (float)r * g.cell * 0.999f;
        closed = best.d[2] <= reach * reach || r >= rmax;
    }
    if (!closed) {  // an outlier: the whole cloud, as the brute-force scan does
        best.init();
        for (int j = 0; j < n; ++j) {
            const float4 q = sorted[j];
            if (__float_as_int(q.w) != self) best.offer(sqdist(me.x, me.y, me.z, q.x, q.y, q.z), 0);
        }
    }
    mean_dist2[self] = ((best.d[0] + best.d[1]) + best.d[2]) / 3.0f;
}

struct GridWorkspace {
    size_t params, cell_id, slot, cell_count, sorted, total;
    explicit GridWorkspace(int n)
    {
        auto up = [](size_t v) { return (v + 255) / 256 * 256; };
        size_t o = 0;
        params = o, o = up(o + sizeof(GridParams));
        cell_id = o, o = up(o + 4 * (size_t)n);
        slot = o, o = up(o + 4 * (size_t)n);
        cell_count = o, o = up(o + 4 * ((size_t)n + 2));
        sorted = o, o = up(o + 16 * (size_t)n);
        total = o;
    }
};

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
struct LaunchLbsMap {
    static int go(int n, const float* p, int m, const float* t, const float* w, int J, const float* vt, const float* info, int Cc,
                  float* od, float* ot, float* oi, int32_t* oidx, float* owgt, hipStream_t st)
    {
        hipLaunchKernelGGL(lbsmap_top_k_kernel<K>, dim3((n + 63) / 64), dim3(64 * KNN_SPLIT), 0, st, n, p, m, t, w, J, vt, info, Cc, od, ot,
                           oi, oidx, owgt);
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

extern "C" int32_t hgs_dist_cuda2(int32_t n, const float* points, float* mean_dist2, void* stream);

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

extern "C" int32_t hgs_smpl_lbsmap_top_k(int32_t n, const float* points, int32_t m, const float* template_points,
                                         const float* lbs_weights, int32_t J, int32_t K, const float* verts_transform,
                                         const float* addition_info, int32_t C, float* out_dist, float* out_transform,
                                         float* out_info, int32_t* out_idx, float* out_wgt, void* stream)
{
    if (n < 0 || m < K || K < 1 || J < 1 || C < 0) return fail_knn("smpl_lbsmap_top_k: need n >= 0, J >= 1, C >= 0 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !lbs_weights || !verts_transform || !out_dist || !out_transform || !out_idx || !out_wgt ||
        ((addition_info != nullptr) != (out_info != nullptr)) || (addition_info && C < 1))
        return fail_knn("smpl_lbsmap_top_k: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("smpl_lbsmap_top_k: template_points must be float-aligned");
    if ((((uintptr_t)verts_transform | (uintptr_t)out_transform) & 15) != 0)
        return fail_knn("smpl_lbsmap_top_k: verts_transform and out_transform must be 16-byte aligned");
    if (int rc = dispatch_k<LaunchLbsMap>(K, n, points, m, template_points, lbs_weights, J, verts_transform, addition_info, C, out_dist,
                                          out_transform, out_info, out_idx, out_wgt, (hipStream_t)stream))
        return rc;
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("smpl_lbsmap_top_k: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_smpl_lbsmap_top_k_backward(int32_t n, int32_t K, const int32_t* idx, const float* wgt,
                                                  const float* dL_dtransform, const float* dL_dinfo, int32_t C,
                                                  float* dL_dverts_transform, float* dL_daddition_info, void* stream)
{
    if (n < 0 || K < 1 || C < 0) return fail_knn("smpl_lbsmap_top_k_backward: bad sizes");
    if (n == 0) return HGS_OK;
    if (!idx || !wgt || (dL_dtransform && !dL_dverts_transform) || (dL_dinfo && !dL_daddition_info))
        return fail_knn("smpl_lbsmap_top_k_backward: null pointer");
    const size_t threads = (size_t)n * K;
    hipLaunchKernelGGL(lbsmap_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, K, idx, wgt,
                       dL_dtransform, dL_dinfo, C, dL_dverts_transform, dL_daddition_info);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("smpl_lbsmap_top_k_backward: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" size_t hgs_dist_cuda2_workspace(int32_t n) { return n >= GRID_MIN_POINTS ? GridWorkspace(n).total : 0; }

extern "C" int32_t hgs_dist_cuda2_ws(int32_t n, const float* points, float* mean_dist2, void* workspace, void* stream)
{
    if (n < GRID_MIN_POINTS || !workspace) return hgs_dist_cuda2(n, points, mean_dist2, stream);
    if (!points || !mean_dist2) return fail_knn("distCUDA2: null pointer");
    if (((uintptr_t)points & 3) != 0) return fail_knn("distCUDA2: points must be float-aligned");
    if (((uintptr_t)workspace & 15) != 0) return fail_knn("distCUDA2: the workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const GridWorkspace ws(n);
    char* base = (char*)workspace;
    GridParams* gp = (GridParams*)(base + ws.params);
    uint32_t* cell_id = (uint32_t*)(base + ws.cell_id);
    uint32_t* slot = (uint32_t*)(base + ws.slot);
    uint32_t* cell_count = (uint32_t*)(base + ws.cell_count);
    float4* sorted = (float4*)(base + ws.sorted);
    const int blocks = (n + 255) / 256;
    if (hipMemsetAsync(cell_count, 0, 4 * ((size_t)n + 2), st) != hipSuccess) return fail_knn("distCUDA2: memset failed");
    hipLaunchKernelGGL(grid_init_kernel, dim3(1), dim3(1), 0, st, gp);
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(blocks), dim3(256), 0, st, n, points, gp);
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(1), 0, st, n, gp, cell_count);
    hipLaunchKernelGGL(grid_count_kernel, dim3(blocks), dim3(256), 0, st, n, points, gp, cell_id, slot, cell_count);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, st, gp, cell_count);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(blocks), dim3(256), 0, st, n, points, cell_id, slot, cell_count, sorted);
    hipLaunchKernelGGL(grid_search_kernel, dim3(blocks), dim3(256), 0, st, n, gp, cell_count, sorted, mean_dist2);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("distCUDA2: kernel launch failed");
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
