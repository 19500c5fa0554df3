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

// The grid search meets its candidates in ANY order: (distance, index) pairs compared lexicographically -- the same total
// order as "ascending index, strict <" above.  A squared distance is >= +0, so its bit pattern orders like the number and the
// pair is ONE 64-bit key, distance bits above the index; inserting is a branch-free pass of compare-and-swap down the sorted
// list (the branchy insertion under a divergent exec mask costs every lane of the wave a long dependent chain whenever one
// lane improves, which with 64 different queries is most trips).
template <int K>
struct BestKeys {
    uint64_t key[K];
    static constexpr uint64_t NEVER = ~0ull;  // (a masked candidate; also what an empty slot holds: distance NaN, index -1)
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) key[k] = NEVER;
    }
    static __device__ __forceinline__ uint64_t make(float dist, int idx) { return ((uint64_t)__float_as_uint(dist) << 32) | (uint32_t)idx; }
    __device__ __forceinline__ bool improves(uint64_t x) const { return x < key[K - 1]; }
    __device__ __forceinline__ void insert(uint64_t x)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool lt = x < key[k];
            const uint64_t lo = lt ? x : key[k], hi = lt ? key[k] : x;
            key[k] = lo, x = hi;
        }
    }
    __device__ __forceinline__ float dist(int k) const { return __uint_as_float((uint32_t)(key[k] >> 32)); }
    __device__ __forceinline__ int index(int k) const { return (int)(uint32_t)key[k]; }
};

// same operation order as pytorch3d's per-dimension accumulation: ((dx^2 + dy^2) + dz^2); built with -ffp-contract=off
__device__ __forceinline__ float sqdist(float px, float py, float pz, float tx, float ty, float tz)
{
    const float dx = px - tx, dy = py - ty, dz = pz - tz;
    return (dx * dx + dy * dy) + dz * dz;
}

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

__device__ __forceinline__ void cell_of(const GridParams& g, float x, float y, float z, int& cx, int& cy, int& cz)
{
    cx = min(max((int)((x - g.minx) * g.inv_cell), 0), g.gx - 1);
    cy = min(max((int)((y - g.miny) * g.inv_cell), 0), g.gy - 1);
    cz = min(max((int)((z - g.minz) * g.inv_cell), 0), g.gz - 1);
}

// A bounding box with an infinite (or NaN) corner -- a cloud with a non-finite coordinate -- has no cell size: the grid then is
// ONE cell (every point clamps into it), which the searches treat like any other degenerate grid: everything is scanned.
__device__ __forceinline__ void grid_box_guard(float* mn, float* ext)
{
    bool finite = true;
    for (int k = 0; k < 3; ++k) finite = finite && fabsf(mn[k]) < 3.0e38f && ext[k] < 3.0e38f;  // (false for NaN too)
    if (!finite)
        for (int k = 0; k < 3; ++k) mn[k] = 0.0f, ext[k] = 0.0f;
}

// ---- K nearest template vertices through a uniform grid over the TEMPLATE ------------------------------------------------
// The brute-force scan below computes n x m distances (110 210 Gaussians x 6 890 SMPL vertices: 0.42 ms per call, and HUGS
// calls it twice per training step, hugs_trimlp.py:318,480).  With a workspace the template is counting-sorted into a grid
// first (one workgroup: bounding box, cell size for about two cells per vertex of the box -- a body fills a few per cent of
// its box --, counts and their scan in LDS, scatter), the queries are counting-sorted by the cell they fall into, and 64
// neighbouring queries at a time test only the vertices of the cells around them (coop_grid_knn below); a query whose K-th
// best distance is not certainly smaller than anything outside that box holds goes to a second pass that scans the whole
// template.  Same fp32 distance expression, candidates compared as (distance, index) pairs: the same K neighbours in the
// same order as the scan, exact ties included.
constexpr int TGRID_MAX_CELLS = 32768, TGRID_MIN_TEMPLATE = 512, TGRID_REACH = 2, TGRID_CELLS_PER_VERTEX = 2, TGRID_MAX_GROUPS = 8;
// the grid pays when the queries crowd the cells (a wave's 64 sorted queries then share a cell or two): at least this many per vertex
constexpr int TGRID_MIN_QUERIES_PER_VERTEX = 4;
inline bool template_grid_pays(int n, int m) { return m >= TGRID_MIN_TEMPLATE && (long long)n >= (long long)TGRID_MIN_QUERIES_PER_VERTEX * m; }
struct TemplateGridWorkspace {
    size_t params, cell_start, sorted, q_count, q_cell_slot, q_order, open_count, open_list, total;
    TemplateGridWorkspace(int n, int m)
    {
        auto up = [](size_t v) { return (v + 255) / 256 * 256; };
        size_t o = 0;
        params = o, o = up(o + sizeof(GridParams));
        cell_start = o, o = up(o + 4 * ((size_t)TGRID_MAX_CELLS + 2));
        sorted = o, o = up(o + 16 * (size_t)m);
        q_count = o, o = up(o + 4 * ((size_t)TGRID_MAX_CELLS + 2));  // queries per template cell, then (in place) its exclusive scan
        q_cell_slot = o, o = up(o + 8 * (size_t)n);                   // (cell, slot inside the cell) of every query
        q_order = o, o = up(o + 4 * (size_t)n);                       // the queries sorted by cell
        open_count = o, o = up(o + 4);                                // queries the box did not close ...
        open_list = o, o = up(o + 4 * (size_t)n);                     // ... and which ones: the second pass scans the template for them
        total = o;
    }
};

__global__ void __launch_bounds__(1024)
template_grid_build_kernel(int m, const float* __restrict__ templ, GridParams* __restrict__ gp, uint32_t* __restrict__ cell_start,
                           float4* __restrict__ sorted, uint32_t* __restrict__ q_count, uint32_t* __restrict__ open_count)
{
    __shared__ uint32_t cnt[TGRID_MAX_CELLS + 1];
    __shared__ uint32_t box[6];
    __shared__ GridParams g;
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 3) box[tid] = 0xFFFFFFFFu, box[3 + tid] = 0u;
    __syncthreads();
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int i = tid; i < m; i += 1024)
        for (int k = 0; k < 3; ++k) {
            const uint32_t key = float_key(templ[3 * (size_t)i + k]);
            lo[k] = min(lo[k], key), hi[k] = max(hi[k], key);
        }
    for (int k = 0; k < 3; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d, 64));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d, 64));
        }
        if (lane == 0) atomicMin(&box[k], lo[k]), atomicMax(&box[3 + k], hi[k]);
    }
    __syncthreads();
    if (tid == 0) {
        float mn[3] = {key_float(box[0]), key_float(box[1]), key_float(box[2])};
        float ext[3];
        for (int k = 0; k < 3; ++k) ext[k] = fmaxf(key_float(box[3 + k]) - mn[k], 0.0f);
        grid_box_guard(mn, ext);
        const float longest = fmaxf(fmaxf(ext[0], ext[1]), fmaxf(ext[2], 1e-30f));
        float cell = longest / 2.0f;
        for (int it = 0; it < 64; ++it) {  // shrink the cell until there are about as many cells as vertices (or the limits are hit)
            const float c = cell * 0.7937005f;  // 2^(-1/3): halves the cell volume
            long long cells = 1;
            bool ok = true;
            for (int k = 0; k < 3; ++k) {
                const long long gk = (long long)floorf(ext[k] / c) + 1;
                ok = ok && gk <= GRID_MAX_DIM;
                cells *= gk;
            }
            if (!ok || cells > (long long)m * TGRID_CELLS_PER_VERTEX || cells > TGRID_MAX_CELLS) break;
            cell = c;
        }
        g.minx = mn[0], g.miny = mn[1], g.minz = mn[2], g.cell = cell, g.inv_cell = 1.0f / cell;
        g.gx = (int)floorf(ext[0] / cell) + 1, g.gy = (int)floorf(ext[1] / cell) + 1, g.gz = (int)floorf(ext[2] / cell) + 1;
        g.cells = g.gx * g.gy * g.gz;
        *gp = g;
    }
    __syncthreads();
    const int cells = g.cells;
    for (int c = tid; c <= cells; c += 1024) cnt[c] = 0u, q_count[c] = 0u;  // (the queries' counters too: no memset launches)
    if (tid == 0) q_count[cells + 1] = 0u, *open_count = 0u;
    __syncthreads();
    for (int i = tid; i < m; i += 1024) {
        int cx, cy, cz;
        cell_of(g, templ[3 * (size_t)i], templ[3 * (size_t)i + 1], templ[3 * (size_t)i + 2], cx, cy, cz);
        atomicAdd(&cnt[(cz * g.gy + cy) * g.gx + cx], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. cells] in place, 8 192 entries per trip
    uint32_t carry = 0;
    for (int base = 0; base <= cells; base += 8192) {
        uint32_t c[8], mine = 0;
        for (int k = 0; k < 8; ++k) {
            const int j = base + tid * 8 + k;
            c[k] = j <= cells ? cnt[j] : 0u, mine += c[k];
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
        uint32_t run = carry + before + incl - mine;
        for (int k = 0; k < 8; ++k) {
            const int j = base + tid * 8 + k;
            if (j <= cells) cnt[j] = run, cell_start[j] = run;
            run += c[k];
        }
        carry += total;
        __syncthreads();
    }
    // scatter: cnt[] now serves as the cells' cursors
    for (int i = tid; i < m; i += 1024) {
        const float x = templ[3 * (size_t)i], y = templ[3 * (size_t)i + 1], z = templ[3 * (size_t)i + 2];
        int cx, cy, cz;
        cell_of(g, x, y, z, cx, cy, cz);
        sorted[atomicAdd(&cnt[(cz * g.gy + cy) * g.gx + cx], 1u)] = make_float4(x, y, z, __int_as_float(i));
    }
}

// The QUERIES are counting-sorted by the template cell they fall into (clamped to the grid), so that a wave's 64 consecutive
// queries are neighbours in space: the wave takes the box of cells its queries span, grown by TGRID_REACH cells, and every
// lane tests every vertex of that box -- a scan with uniform (scalar-cache) vertex loads like the full one, over a few
// hundred vertices instead of all of them.  Cells that are neighbours in x are neighbours in the sorted template, so a row of
// the box is ONE contiguous range of vertices.  A lane whose K-th best distance is not certainly inside what the box covers
// for it (TGRID_REACH cell sizes around its own cell) starts over and scans the whole template, as do all lanes of a wave
// whose box is most of the grid anyway.
__global__ void __launch_bounds__(256)
query_count_kernel(int n, const float* __restrict__ points, const GridParams* __restrict__ gp, uint32_t* __restrict__ q_count,
                   uint2* __restrict__ q_cell_slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const GridParams g = *gp;
    int cx, cy, cz;
    cell_of(g, points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2], cx, cy, cz);
    const uint32_t c = (uint32_t)((cz * g.gy + cy) * g.gx + cx);
    q_cell_slot[i] = make_uint2(c, atomicAdd(&q_count[c], 1u));
}
__global__ void __launch_bounds__(1024) query_scan_kernel(const GridParams* __restrict__ gp, uint32_t* __restrict__ q_count)
{
    __shared__ uint32_t wsum[16];
    const int cells = gp->cells + 1, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < cells; base += 8192) {
        uint32_t c[8], mine = 0;
        for (int k = 0; k < 8; ++k) {
            const int j = base + threadIdx.x * 8 + k;
            c[k] = j < cells ? q_count[j] : 0u, mine += c[k];
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
            if (j < cells) q_count[j] = run;
            run += c[k];
        }
        carry += total;
    }
}
__global__ void __launch_bounds__(256)
query_scatter_kernel(int n, const uint2* __restrict__ q_cell_slot, const uint32_t* __restrict__ q_start, uint32_t* __restrict__ q_order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 cs = q_cell_slot[i];
    q_order[q_start[cs.x] + cs.y] = (uint32_t)i;
}

// what the search kernels below get when the template has been put on a grid (all null: the scan)
struct TemplateGrid {
    const GridParams* params; const uint32_t* cell_start; const float4* sorted; const uint32_t* q_order;
    uint32_t* open_count; uint32_t* open_list;
    int pass;  // 1: the box search over the sorted queries, 2: the scan for the queries pass 1 left open
};
constexpr int KNN_WAVES = 4;  // waves per workgroup of the search kernels
struct WaveStage { float4 v[64]; uint32_t row_end[64], row_src[64]; };

// A workgroup = 64 consecutive queries of the sorted order (one per lane, `valid` lanes hold one), the same in each of its
// four waves.  The waves work through GROUPS of the lanes: the first pending lane leads, the pending lanes whose cell is the
// leader's or touches it join, and the group tests every vertex of the box of cells it spans grown by TGRID_REACH (the other
// lanes idle: 64 sorted queries are one group almost always, two where they cross into the next row of cells, which may be
// far away -- one box over both would be most of the grid).  The four waves share a box's vertices chunk by chunk (one wave
// per 64 queries leaves a SIMD with one or two waves and every dependent-instruction latency exposed: 0.19 us per vertex).
// Candidate vertices are offered as (distance, index) pairs in any order.  Returns whether the lane was in a group whose box
// was searched; `best` then holds this WAVE's share of the candidates.
template <int K>
__device__ __forceinline__ bool coop_grid_knn(const TemplateGrid& tg, const GridParams& g, bool valid, float px, float py, float pz, BestKeys<K>& best)
{
    best.init();
    int cx, cy, cz;
    cell_of(g, px, py, pz, cx, cy, cz);
    // The vertices of a group's box reach the lanes through the wave's own LDS stage, 64 per chunk: a row of the box (cells
    // that are neighbours in x) is ONE range of the sorted template, lane r fetches row r's range (all rows: one memory
    // latency), a prefix sum lays the ranges end to end, and lane l of a chunk looks up which row its vertex is in (a binary
    // search over the <= 64 row ends in LDS) and loads it -- 64 vertices per load instruction, the next chunk's in flight while
    // this one is tested.  (Dependent scalar loads of four vertices a trip cost a full L2-miss latency per trip: 0.7 us.)
    __shared__ WaveStage stage_all[KNN_WAVES];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveStage& st = stage_all[w];
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    bool searched = false, pending = valid;
    for (int round = 0; round < TGRID_MAX_GROUPS; ++round) {
        const uint64_t todo = __builtin_amdgcn_ballot_w64(pending);
        if (todo == 0ull) break;
        const int leader = __builtin_ctzll(todo);
        const int Lx = __builtin_amdgcn_readlane(cx, leader), Ly = __builtin_amdgcn_readlane(cy, leader), Lz = __builtin_amdgcn_readlane(cz, leader);
        const bool in = pending && abs(cx - Lx) <= 1 && abs(cy - Ly) <= 1 && abs(cz - Lz) <= 1;
        pending = pending && !in;
        // the group's span per axis is a sub-range of leader-1 .. leader+1
        auto some = [&](bool c) { return __builtin_amdgcn_ballot_w64(in && c) != 0ull; };
        const int x0 = max((some(cx < Lx) ? Lx - 1 : Lx) - TGRID_REACH, 0), x1 = min((some(cx > Lx) ? Lx + 1 : Lx) + TGRID_REACH, g.gx - 1);
        const int y0 = max((some(cy < Ly) ? Ly - 1 : Ly) - TGRID_REACH, 0), y1 = min((some(cy > Ly) ? Ly + 1 : Ly) + TGRID_REACH, g.gy - 1);
        const int z0 = max((some(cz < Lz) ? Lz - 1 : Lz) - TGRID_REACH, 0), z1 = min((some(cz > Lz) ? Lz + 1 : Lz) + TGRID_REACH, g.gz - 1);
        const int ny = y1 - y0 + 1, rows = ny * (z1 - z0 + 1);  // (<= 7 x 7)
        if ((x1 - x0 + 1) * rows * 2 > g.cells) continue;  // most of the grid: the scan for this group
        uint32_t first = 0, len = 0;
        if (lane < rows) {
            const int rz = lane / ny, row = ((z0 + rz) * g.gy + y0 + (lane - rz * ny)) * g.gx;
            first = tg.cell_start[row + x0], len = tg.cell_start[row + x1 + 1] - first;
        }
        uint32_t end = len;  // inclusive prefix sum: where the row ends in the box's list of vertices
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)end, d, 64);
            if (lane >= d) end += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
        wave_sync();  // (the previous group's last reads of the stage are done)
        st.row_end[lane] = end, st.row_src[lane] = first - (end - len);  // vertex f of the list = sorted[row_src[row] + f]
        wave_sync();
        auto fetch = [&](uint32_t c) {
            const uint32_t f = c + (uint32_t)lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < total) {
                int r = 0;  // the first row that ends beyond f (empty rows end where their predecessor does: skipped)
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1)
                    if (st.row_end[r + step - 1] <= f) r += step;
                v = tg.sorted[st.row_src[r] + f];
            }
            return v;
        };
        searched = searched || in;
        float4 next = fetch(64u * (uint32_t)w);  // this wave's chunks: w, w + 4, ...
        for (uint32_t c = 64u * (uint32_t)w; c < total; c += 64u * KNN_WAVES) {
            wave_sync();
            st.v[lane] = next;
            wave_sync();
            if (c + 64u * KNN_WAVES < total) next = fetch(c + 64u * KNN_WAVES);
            const uint32_t count = min(64u, total - c);  // (uniform)
            for (uint32_t k = 0; k < count; k += 4u) {
                const float4 a = st.v[k], b = st.v[k + 1], cc = st.v[k + 2], dd = st.v[k + 3];  // same address in every lane: broadcasts
                const uint32_t left = count - k;
                const float e1 = sqdist(px, py, pz, b.x, b.y, b.z), e2 = sqdist(px, py, pz, cc.x, cc.y, cc.z), e3 = sqdist(px, py, pz, dd.x, dd.y, dd.z);
                const float d0 = sqdist(px, py, pz, a.x, a.y, a.z);
                const uint64_t never = BestKeys<K>::NEVER;
                const uint64_t k0 = BestKeys<K>::make(d0, __float_as_int(a.w));
                const uint64_t k1 = left > 1u ? BestKeys<K>::make(e1, __float_as_int(b.w)) : never;
                const uint64_t k2 = left > 2u ? BestKeys<K>::make(e2, __float_as_int(cc.w)) : never;
                const uint64_t k3 = left > 3u ? BestKeys<K>::make(e3, __float_as_int(dd.w)) : never;
                // one wave-level test for "nobody improves", the usual case
                const bool any = in && (best.improves(k0) || best.improves(k1) || best.improves(k2) || best.improves(k3));
                if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;
                if (in) best.insert(k0), best.insert(k1), best.insert(k2), best.insert(k3);
            }
        }
    }
    return searched;
}

constexpr int KNN_SPLIT = KNN_WAVES;  // waves per workgroup = template segments

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
__device__ __forceinline__ bool workgroup_knn(int pc, const float* __restrict__ points, int m, const float* __restrict__ templ,
                                              Best<K>& best)
{
    __shared__ float sh_d[KNN_SPLIT - 1][K][64];
    __shared__ int sh_i[KNN_SPLIT - 1][K][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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

// true in the threads that hold a query's result (`i`).  Without a grid: the workgroup's scan for queries blockIdx.x*64+lane.
// With one, pass 1: a workgroup = 64 consecutive queries of the order sorted by template cell, searched in their box, the
// four waves' shares merged by wave 0; the lanes the box did not close put their query on the open list (one atomic per
// workgroup) and pass 2 -- launched for the worst case, workgroups beyond the list's end leave at once -- scans the template
// for those.
template <int K>
__device__ __forceinline__ bool search(int n, const float* __restrict__ points, int m, const float* __restrict__ templ,
                                       const TemplateGrid& tg, Best<K>& best, int& i)
{
    const int lane = threadIdx.x & 63;
    if (tg.params && tg.pass == 1) {  // (uniform)
        __shared__ uint64_t sh_key[KNN_WAVES - 1][K][64];
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int sidx = blockIdx.x * 64 + lane;
        const bool valid = sidx < n;
        i = (int)tg.q_order[valid ? sidx : n - 1];
        const float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
        const GridParams g = *tg.params;
        BestKeys<K> keys;
        const bool searched = coop_grid_knn<K>(tg, g, valid, px, py, pz, keys);
        if (w > 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) sh_key[w - 1][k][lane] = keys.key[k];
        }
        __syncthreads();
        if (w > 0) return false;
#pragma unroll
        for (int q = 0; q < KNN_WAVES - 1; ++q)
#pragma unroll
            for (int k = 0; k < K; ++k) keys.insert(sh_key[q][k][lane]);
#pragma unroll
        for (int k = 0; k < K; ++k) best.d[k] = keys.dist(k), best.i[k] = keys.index(k);
        // the box covers at least TGRID_REACH cell sizes around the lane's own cell (less where the grid ends: nothing lies
        // beyond): anything outside is at least that far away (a little less: a cell index is an fp32 floor).  Strictly
        // smaller: a vertex at exactly that distance could still win a tie with a lower index.  (A NaN query: nothing is
        // ever closer than +inf, the scan's path for it.)
        const float reach = (float)TGRID_REACH * g.cell * 0.999f;
        const bool closed = searched && px == px && py == py && pz == pz && best.d[K - 1] < reach * reach;
        const bool open = valid && !closed;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(open);
        if (mask != 0ull) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(tg.open_count, (uint32_t)__builtin_popcountll(mask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (open) tg.open_list[base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
        return valid && closed;
    }
    int q = blockIdx.x * 64 + lane, count = n;
    if (tg.params) {
        count = (int)*(const volatile uint32_t*)tg.open_count;
        if ((int)(blockIdx.x * 64) >= count) return false;  // (uniform)
        i = (int)tg.open_list[min(q, count - 1)];
    } else {
        i = min(q, n - 1);  // every lane scans (wave-uniform loads); only valid lanes store
    }
    return workgroup_knn<K>(i, points, m, templ, best) && q < count;
}

template <int K>
__global__ void __launch_bounds__(64 * KNN_SPLIT)
knn_kernel(int n, const float* __restrict__ points, int m, const float* __restrict__ templ, float* __restrict__ dists,
           int64_t* __restrict__ idx, TemplateGrid tg)
{
    Best<K> best;
    int i;
    if (!search<K>(n, points, m, templ, tg, best, i)) return;
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
                       float* __restrict__ out_weights, TemplateGrid tg)
{
    Best<K> best;
    int i;
    if (!search<K>(n, points, m, templ, tg, best, i)) return;
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
                    float* __restrict__ out_info, int32_t* __restrict__ out_idx, float* __restrict__ out_wgt, TemplateGrid tg)
{
    Best<K> best;
    int i;
    if (!search<K>(n, points, m, templ, tg, best, i)) return;
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
    const int q = blockIdx.x * 64 + (threadIdx.x & 63), i = min(q, n - 1);
    if (!workgroup_knn<3, true>(i, points, n, points, best) || q >= n) return;
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
    float mn[3] = {key_float(g->lo[0]), key_float(g->lo[1]), key_float(g->lo[2])};
    float ext[3];
    for (int k = 0; k < 3; ++k) ext[k] = fmaxf(key_float(g->hi[k]) - mn[k], 0.0f);
    grid_box_guard(mn, ext);
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

// launches `kernel` for the search: once without a grid, else the box pass and the scan of what it left open
template <typename Kernel, typename... A>
inline void launch_search(Kernel kernel, int n, TemplateGrid tg, hipStream_t st, A... a)
{
    const int scan_blocks = (n + 63) / 64;
    if (!tg.params) {
        hipLaunchKernelGGL(kernel, dim3(scan_blocks), dim3(64 * KNN_SPLIT), 0, st, a..., tg);
        return;
    }
    tg.pass = 1;
    hipLaunchKernelGGL(kernel, dim3(scan_blocks), dim3(64 * KNN_SPLIT), 0, st, a..., tg);
    tg.pass = 2;
    hipLaunchKernelGGL(kernel, dim3(scan_blocks), dim3(64 * KNN_SPLIT), 0, st, a..., tg);
}

// puts the template on a grid in `workspace` (hgs_knn_workspace(m) bytes) when that pays: enough vertices, a workspace given
TemplateGrid build_template_grid(int n, const float* points, int m, const float* templ, void* workspace, hipStream_t st)
{
    if (!workspace || !template_grid_pays(n, m)) return TemplateGrid{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    const TemplateGridWorkspace ws(n, m);
    char* base = (char*)workspace;
    GridParams* gp = (GridParams*)(base + ws.params);
    uint32_t* cell_start = (uint32_t*)(base + ws.cell_start);
    float4* sorted = (float4*)(base + ws.sorted);
    uint32_t* q_count = (uint32_t*)(base + ws.q_count);
    uint2* q_cell_slot = (uint2*)(base + ws.q_cell_slot);
    uint32_t* q_order = (uint32_t*)(base + ws.q_order);
    uint32_t* open_count = (uint32_t*)(base + ws.open_count);
    uint32_t* open_list = (uint32_t*)(base + ws.open_list);
    hipLaunchKernelGGL(template_grid_build_kernel, dim3(1), dim3(1024), 0, st, m, templ, gp, cell_start, sorted, q_count, open_count);
    hipLaunchKernelGGL(query_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, points, gp, q_count, q_cell_slot);
    hipLaunchKernelGGL(query_scan_kernel, dim3(1), dim3(1024), 0, st, gp, q_count);
    hipLaunchKernelGGL(query_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, q_cell_slot, q_count, q_order);
    return TemplateGrid{gp, cell_start, sorted, q_order, open_count, open_list, 1};
}

template <int K>
struct LaunchKnn {
    static int go(int n, const float* p, int m, const float* t, float* d, int64_t* idx, TemplateGrid tg, hipStream_t st)
    {
        launch_search(knn_kernel<K>, n, tg, st, n, p, m, t, d, idx);
        return HGS_OK;
    }
};
template <int K>
struct LaunchLbsMap {
    static int go(int n, const float* p, int m, const float* t, const float* w, int J, const float* vt, const float* info, int Cc,
                  float* od, float* ot, float* oi, int32_t* oidx, float* owgt, TemplateGrid tg, hipStream_t st)
    {
        launch_search(lbsmap_top_k_kernel<K>, n, tg, st, n, p, m, t, w, J, vt, info, Cc, od, ot, oi, oidx, owgt);
        return HGS_OK;
    }
};
template <int K>
struct LaunchLbs {
    static int go(int n, const float* p, int m, const float* t, const float* w, int J, float* od, float* ow, TemplateGrid tg, hipStream_t st)
    {
        launch_search(lbsweight_top_k_kernel<K>, n, tg, st, n, p, m, t, w, J, od, ow);
        return HGS_OK;
    }
};

}  // namespace

extern "C" int32_t hgs_dist_cuda2(int32_t n, const float* points, float* mean_dist2, void* stream);

extern "C" size_t hgs_knn_workspace(int32_t n, int32_t m) { return template_grid_pays(n, m) ? TemplateGridWorkspace(n, m).total : 0; }

extern "C" int32_t hgs_knn_points_ws(int32_t n, const float* points, int32_t m, const float* template_points, int32_t K,
                                     float* dists, int64_t* idx, void* workspace, void* stream);
extern "C" int32_t hgs_knn_points(int32_t n, const float* points, int32_t m, const float* template_points, int32_t K,
                                  float* dists, int64_t* idx, void* stream)
{
    return hgs_knn_points_ws(n, points, m, template_points, K, dists, idx, nullptr, stream);
}

extern "C" int32_t hgs_knn_points_ws(int32_t n, const float* points, int32_t m, const float* template_points, int32_t K,
                                     float* dists, int64_t* idx, void* workspace, void* stream)
{
    if (n < 0 || m < K || K < 1) return fail_knn("knn_points: need n >= 0 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !dists || !idx) return fail_knn("knn_points: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("knn_points: template_points must be float-aligned");
    if (((uintptr_t)workspace & 15) != 0) return fail_knn("knn_points: the workspace must be 16-byte aligned");
    const TemplateGrid tg = build_template_grid(n, points, m, template_points, workspace, (hipStream_t)stream);
    if (int rc = dispatch_k<LaunchKnn>(K, n, points, m, template_points, dists, idx, tg, (hipStream_t)stream)) return rc;
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("knn_points: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_smpl_lbsweight_top_k_ws(int32_t n, const float* points, int32_t m, const float* template_points,
                                               const float* lbs_weights, int32_t J, int32_t K, float* out_dist,
                                               float* out_weights, void* workspace, void* stream);
extern "C" int32_t hgs_smpl_lbsweight_top_k(int32_t n, const float* points, int32_t m, const float* template_points,
                                            const float* lbs_weights, int32_t J, int32_t K, float* out_dist,
                                            float* out_weights, void* stream)
{
    return hgs_smpl_lbsweight_top_k_ws(n, points, m, template_points, lbs_weights, J, K, out_dist, out_weights, nullptr, stream);
}

extern "C" int32_t hgs_smpl_lbsweight_top_k_ws(int32_t n, const float* points, int32_t m, const float* template_points,
                                               const float* lbs_weights, int32_t J, int32_t K, float* out_dist,
                                               float* out_weights, void* workspace, void* stream)
{
    if (n < 0 || m < K || K < 1 || J < 1) return fail_knn("smpl_lbsweight_top_k: need n >= 0, J >= 1 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !lbs_weights || !out_dist || !out_weights) return fail_knn("smpl_lbsweight_top_k: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("smpl_lbsweight_top_k: template_points must be float-aligned");
    if (((uintptr_t)workspace & 15) != 0) return fail_knn("smpl_lbsweight_top_k: the workspace must be 16-byte aligned");
    const TemplateGrid tg = build_template_grid(n, points, m, template_points, workspace, (hipStream_t)stream);
    if (int rc = dispatch_k<LaunchLbs>(K, n, points, m, template_points, lbs_weights, J, out_dist, out_weights, tg, (hipStream_t)stream)) return rc;
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("smpl_lbsweight_top_k: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_smpl_lbsmap_top_k(int32_t n, const float* points, int32_t m, const float* template_points,
                                         const float* lbs_weights, int32_t J, int32_t K, const float* verts_transform,
                                         const float* addition_info, int32_t C, float* out_dist, float* out_transform,
                                         float* out_info, int32_t* out_idx, float* out_wgt, void* workspace, void* stream)
{
    if (n < 0 || m < K || K < 1 || J < 1 || C < 0) return fail_knn("smpl_lbsmap_top_k: need n >= 0, J >= 1, C >= 0 and 1 <= K <= m");
    if (n == 0) return HGS_OK;
    if (!points || !template_points || !lbs_weights || !verts_transform || !out_dist || !out_transform || !out_idx || !out_wgt ||
        ((addition_info != nullptr) != (out_info != nullptr)) || (addition_info && C < 1))
        return fail_knn("smpl_lbsmap_top_k: null pointer");
    if (((uintptr_t)template_points & 3) != 0) return fail_knn("smpl_lbsmap_top_k: template_points must be float-aligned");
    if ((((uintptr_t)verts_transform | (uintptr_t)out_transform) & 15) != 0)
        return fail_knn("smpl_lbsmap_top_k: verts_transform and out_transform must be 16-byte aligned");
    if (((uintptr_t)workspace & 15) != 0) return fail_knn("smpl_lbsmap_top_k: the workspace must be 16-byte aligned");
    const TemplateGrid tg = build_template_grid(n, points, m, template_points, workspace, (hipStream_t)stream);
    if (int rc = dispatch_k<LaunchLbsMap>(K, n, points, m, template_points, lbs_weights, J, verts_transform, addition_info, C, out_dist,
                                          out_transform, out_info, out_idx, out_wgt, tg, (hipStream_t)stream))
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
