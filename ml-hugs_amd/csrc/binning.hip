// Tile binning (scan, key emission, sort, tile ranges), restructured for MI355X:
//
//   K1 (preprocess.hip) counts, with fire-and-forget atomics, how many Gaussians touch every tile
//   tile_scan   one workgroup: exclusive scan of the tile counts -> ranges[tile] = [start,end), N = total
//   emit        wave-balanced: every (Gaussian, tile) pair takes a slot of its tile's segment with a returning
//               atomic and stores (quad coverage mask << 28 | Gaussian index) there          -- a bucket scatter
//   tile_sort   one workgroup per tile: bitonic sort of the segment in LDS by (fp32 depth bits, Gaussian index)
//   bitmaps / compact   per-quad bitmaps over the sorted list and the compacted lists the blend kernels stream
//
// The result is, for every tile, exactly the order a stable sort of the 64-bit keys (tile << 32 | depth bits)
// produces (ties: ascending Gaussian index) -- bit-identical to the oracle's sorted list -- without ever moving a
// 64-bit key through a multi-pass global radix sort: ~36 B per entry of HBM traffic instead of ~160 B, and 6 kernel
// launches instead of 20.  The slot order inside a segment before sorting is arbitrary (atomics); the per-tile
// sort is on a total order, so the output is deterministic.
#include "hgs_common.h"

namespace hgs {

// ---------------------------------------------------------------------------------------------
// wave / block scan helpers (wave = 64 lanes)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// inclusive scan over a 256-thread block of one value per thread; returns the inclusive value and the
// block total through `total`. `wsum` is a 4-entry LDS array.
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wsum, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t s = wsum[k];
        if (k < w) base += s;
    }
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return inc + base;
}

constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;  // 1024 elements per block

__global__ void __launch_bounds__(256) scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ block_sums, int n)
{
    __shared__ uint32_t wsum[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    block_inclusive_scan(s, wsum, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of block_sums[0..m) in place
__global__ void __launch_bounds__(256) scan_sums_kernel(uint32_t* __restrict__ block_sums, int m)
{
    __shared__ uint32_t wsum[4];
    uint32_t carry = 0;
    for (int base = 0; base < m; base += 256) {
        int i = base + threadIdx.x;
        uint32_t v = i < m ? block_sums[i] : 0;
        uint32_t total;
        uint32_t inc = block_inclusive_scan(v, wsum, total);
        if (i < m) block_sums[i] = carry + inc - v;
        carry += total;
    }
}

__global__ void __launch_bounds__(256) scan_apply_kernel(const uint32_t* in, const uint32_t* __restrict__ block_sums,
                                                         uint32_t* out, int n)  // in may alias out
{
    __shared__ uint32_t wsum[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t total;
    uint32_t inc = block_inclusive_scan(s, wsum, total);
    uint32_t run = block_sums[blockIdx.x] + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
}

void launch_scan_inclusive(const uint32_t* in, uint32_t* out, uint32_t* tmp, int n, hipStream_t st)
{
    if (n <= 0) return;
    int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(256), 0, st, in, tmp, n);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, tmp, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, st, in, tmp, out, n);
}

// ---------------------------------------------------------------------------------------------
// tile_scan: single workgroup.  Integrates K1's 2-D difference array into per-tile counts (row pass, column
// pass; in LDS when the grid fits, in place in global memory otherwise), then scans the counts in tile order.
// ranges[t] = (0,0) for an empty tile (as the oracle leaves them).
constexpr int TILE_SCAN_LDS_CELLS = 36 * 1024;  // 144 KB of the CU's 160 KB

__global__ void __launch_bounds__(1024)
tile_scan_kernel(uint32_t* __restrict__ diff, int gx, int gy, uint2* __restrict__ ranges, uint32_t* __restrict__ cursor,
                 uint32_t* __restrict__ n_total)
{
    extern __shared__ uint32_t grid_lds[];
    __shared__ uint32_t wsum[16];
    const int pitch = gx + 1, cells = pitch * (gy + 1), num_tiles = gx * gy;
    const bool in_lds = cells <= TILE_SCAN_LDS_CELLS;
    uint32_t* g = in_lds ? grid_lds : diff;
    if (in_lds) {
        for (int i = threadIdx.x; i < cells; i += 1024) g[i] = diff[i];
        __syncthreads();
    }
    for (int y = threadIdx.x; y <= gy; y += 1024) {  // prefix along x
        uint32_t acc = 0;
        for (int x = 0; x <= gx; ++x) acc += g[y * pitch + x], g[y * pitch + x] = acc;
    }
    __syncthreads();
    for (int x = threadIdx.x; x <= gx; x += 1024) {  // prefix along y
        uint32_t acc = 0;
        for (int y = 0; y <= gy; ++y) acc += g[y * pitch + x], g[y * pitch + x] = acc;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < num_tiles; base += 1024) {
        const int t = base + threadIdx.x;
        const uint32_t c = t < num_tiles ? g[(t / gx) * pitch + (t % gx)] : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t v = wsum[k];
            if (k < w) before += v;
            total += v;
        }
        __syncthreads();
        if (t < num_tiles) {
            const uint32_t start = carry + before + inc - c;
            ranges[t] = c ? make_uint2(start, start + c) : make_uint2(0u, 0u);
            cursor[t] = start;
        }
        carry += total;
    }
    if (threadIdx.x == 0) *n_total = carry;
}

void launch_tile_scan(uint32_t* tile_count_diff, int gx, int gy, uint2* ranges, uint32_t* cursor, uint32_t* n_total,
                      hipStream_t st)
{
    const int cells = (gx + 1) * (gy + 1);
    const size_t lds = cells <= TILE_SCAN_LDS_CELLS ? sizeof(uint32_t) * (size_t)cells : 0;
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), lds, st, tile_count_diff, gx, gy, ranges, cursor, n_total);
}

// ---------------------------------------------------------------------------------------------
// emit: bucket scatter, wave-balanced, plus a per-entry quad coverage mask.
//
// A wave owns 64 consecutive Gaussians; their tile counts are prefix-summed in registers and the wave's
// (Gaussian, tile) pairs are dealt to lanes 64 at a time (each pair finds its Gaussian by a 6-step search over
// the wave's prefix sums), so a single huge splat no longer serialises one lane.
//
// The value's top 4 bits carry a coverage mask: bit q is set when the splat can reach alpha >= 1/255 on
// some pixel of the tile's 8x8 quad q (q = qx + 2 qy).  It is CONSERVATIVE (may be set needlessly, never
// missing): the blend kernels skip a (quad, splat) pair whose bit is clear without touching a VGPR.
// max over the pixel-centre rectangle [x0,x0+7] x [y0,y0+7] of  A dx^2 + B dx dy + C dy^2  (concave)
__device__ __forceinline__ float max_power_in_quad(float sx, float sy, float A, float B, float C, float x0, float y0)
{
    const float dxl = sx - (x0 + 7.0f), dxh = sx - x0, dyl = sy - (y0 + 7.0f), dyh = sy - y0;
    if (dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f) return 0.0f;  // centre inside the quad
    const float hA = -0.5f / A, hC = -0.5f / C;  // vertex of the 1-D restriction: d* = -B d_other / (2 A|C)
    float best = -3.0e38f;
    {
        float dy = fminf(dyh, fmaxf(dyl, B * dxl * hC));
        best = fmaxf(best, A * dxl * dxl + (B * dxl + C * dy) * dy);
        dy = fminf(dyh, fmaxf(dyl, B * dxh * hC));
        best = fmaxf(best, A * dxh * dxh + (B * dxh + C * dy) * dy);
        float dx = fminf(dxh, fmaxf(dxl, B * dyl * hA));
        best = fmaxf(best, C * dyl * dyl + (B * dyl + A * dx) * dx);
        dx = fminf(dxh, fmaxf(dxl, B * dyh * hA));
        best = fmaxf(best, C * dyh * dyh + (B * dyh + A * dx) * dx);
    }
    return best;
}

__global__ void __launch_bounds__(256)
emit_kernel(int P, Camera cam, const Splat* __restrict__ splats, uint32_t* __restrict__ cursor,
            uint32_t* __restrict__ values)
{
    __shared__ float4 stage[4][64][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + w) * 64;
    if (g0 >= P) return;  // whole wave
    const int g = g0 + lane;

    uint32_t cnt = 0;
    float4 r0 = make_float4(0.f, 0.f, -1.f, 0.f), r1 = make_float4(-1.f, 3.0e38f, 0.f, 0.f), r2 = make_float4(0.f, 0.f, 1.f, 0.f);
    if (g < P) {
        const float4 tail = reinterpret_cast<const float4*>(splats + g)[2];
        const int radius = __float_as_int(tail.z);
        if (radius > 0) {
            const float4 head = reinterpret_cast<const float4*>(splats + g)[0];
            const float4 mid = reinterpret_cast<const float4*>(splats + g)[1];
            const float px = head.x, py = head.y, radf = (float)radius;
            // identical expressions to the preprocess kernel => identical rectangle
            const int minx = (int)fminf((float)cam.gx, fmaxf(0.0f, (px - radf) / 16.0f));
            const int maxx = (int)fminf((float)cam.gx, fmaxf(0.0f, (px + radf + 15.0f) / 16.0f));
            const int miny = (int)fminf((float)cam.gy, fmaxf(0.0f, (py - radf) / 16.0f));
            const int maxy = (int)fminf((float)cam.gy, fmaxf(0.0f, (py + radf + 15.0f) / 16.0f));
            cnt = (uint32_t)((maxx - minx) * (maxy - miny));
            // contributes iff opacity * exp(power) >= 1/255  <=>  power >= -(ln 255 + ln opacity); 0.05 of slack
            // covers the blend kernels' rounding (and makes the mask a strict superset)
            const float thr = -(5.5412635f + __logf(mid.y)) - 0.05f;
            r0 = make_float4(px, py, head.z, head.w);
            r1 = make_float4(mid.x, thr, 0.f, 0.f);
            r2 = make_float4(__int_as_float(minx), __int_as_float(miny), __int_as_float(maxx - minx), 0.f);
        }
    }
    const uint32_t incl = wave_inclusive_scan(cnt);
    r1.w = __uint_as_float(incl - cnt);
    stage[w][lane][0] = r0, stage[w][lane][1] = r1, stage[w][lane][2] = r2;
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);

    // four rounds of 64 pairs per trip: the four returning atomics of a lane are independent and in flight together
    for (uint32_t s0 = 0; s0 < total; s0 += 256) {
        uint32_t tile_id[4], val[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = s0 + u * 64 + lane;
            // owner = smallest lane whose inclusive count exceeds s (all lanes take part in the shuffles)
            int lo = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t v = (uint32_t)__shfl((int)incl, lo + step - 1, 64);
                if (v <= s) lo += step;
            }
            valid[u] = s < total;
            tile_id[u] = 0, val[u] = 0;
            if (valid[u]) {
                const float4 a = stage[w][lo][0], b = stage[w][lo][1], c = stage[w][lo][2];
                const uint32_t k = s - __float_as_uint(b.w);
                const uint32_t wdt = (uint32_t)__float_as_int(c.z);
                const uint32_t ry = k / wdt, rx = k - ry * wdt;
                const int tx = __float_as_int(c.x) + (int)rx, ty = __float_as_int(c.y) + (int)ry;
                const float A = a.z, B = a.w, C = b.x, thr = b.y;
                uint32_t mask = 0xFu;
                if (A < 0.0f && C < 0.0f && 4.0f * A * C - B * B > 0.0f) {
                    mask = 0;
                    const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (max_power_in_quad(a.x, a.y, A, B, C, x0 + (float)((q & 1) * 8), y0 + (float)((q >> 1) * 8)) >= thr)
                            mask |= 1u << q;
                }
                tile_id[u] = (uint32_t)(ty * cam.gx + tx);
                val[u] = (mask << GID_BITS) | (uint32_t)(g0 + lo);
            }
        }
        uint32_t slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) slot[u] = valid[u] ? atomicAdd(&cursor[tile_id[u]], 1u) : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (valid[u]) values[slot[u]] = val[u];
    }
}

void launch_emit(int P, const Camera& cam, const Splat* splats, uint32_t* cursor, uint32_t* values, hipStream_t st)
{
    hipLaunchKernelGGL(emit_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, cam, splats, cursor, values);
}

// ---------------------------------------------------------------------------------------------
// tile_sort: one workgroup per tile; CAP = LDS capacity in entries.  Kernel<CAP_SMALL> sorts the tiles with
// n <= CAP_SMALL, kernel<CAP_LARGE> those with CAP_SMALL < n <= CAP_LARGE and -- by brute-force ranking through
// global scratch, slow but only for absurdly dense tiles -- everything longer.
// Sort key: (depth bits << 32) | (gaussian << 4) | mask: depth first, then Gaussian index (the mask rides along).
// Output: list[i] = (pos1 << 32) | (mask << 28 | gaussian), pos1 = 1-based position inside the tile.
constexpr int SORT_CAP_SMALL = 1024, SORT_CAP_LARGE = 8192;

__device__ __forceinline__ uint64_t sort_key(const Splat* __restrict__ splats, uint32_t v)
{
    const uint32_t gid = v & GID_MASK;
    const uint32_t dbits = __float_as_uint(reinterpret_cast<const float*>(splats + gid)[9]);
    return ((uint64_t)dbits << 32) | (uint64_t)((gid << 4) | (v >> GID_BITS));
}
__device__ __forceinline__ uint64_t list_entry(uint64_t key, uint32_t pos1)
{
    const uint32_t low = (uint32_t)key;
    return ((uint64_t)pos1 << 32) | (uint64_t)(((low & 15u) << GID_BITS) | (low >> 4));
}

template <int CAP, bool IS_LARGE>
__global__ void __launch_bounds__(256)
tile_sort_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ values, const Splat* __restrict__ splats,
                 uint64_t* __restrict__ list, uint64_t* __restrict__ scratch)
{
    __shared__ uint64_t sh[CAP];
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t s = rg.x, n = rg.y - rg.x;
    if (n == 0) return;
    if (IS_LARGE ? n <= (uint32_t)SORT_CAP_SMALL : n > (uint32_t)CAP) return;  // the other launch's tile
    if (n <= (uint32_t)CAP) {
        uint32_t m = 2;
        while (m < n) m <<= 1;
        for (uint32_t i = threadIdx.x; i < m; i += 256) sh[i] = i < n ? sort_key(splats, values[s + i]) : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t c = threadIdx.x; c < (m >> 1); c += 256) {
                    const uint32_t l = ((c & ~(j - 1u)) << 1) | (c & (j - 1u)), r = l | j;
                    const uint64_t a = sh[l], b = sh[r];
                    if ((a > b) == ((l & k) == 0u)) sh[l] = b, sh[r] = a;
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < n; i += 256) list[s + i] = list_entry(sh[i], i + 1u);
    } else {
        // n > CAP_LARGE: rank every key against all others (keys are distinct: they embed the Gaussian index)
        for (uint32_t i = threadIdx.x; i < n; i += 256) scratch[s + i] = sort_key(splats, values[s + i]);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint64_t ki = scratch[s + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) rank += scratch[s + j] < ki ? 1u : 0u;
            list[s + rank] = list_entry(ki, rank + 1u);
        }
    }
}

void launch_tile_sort(const uint2* ranges, int num_tiles, const uint32_t* values, const Splat* splats, uint64_t* list,
                      uint64_t* scratch, hipStream_t st)
{
    hipLaunchKernelGGL((tile_sort_kernel<SORT_CAP_SMALL, false>), dim3(num_tiles), dim3(256), 0, st, ranges, values, splats,
                       list, scratch);
    hipLaunchKernelGGL((tile_sort_kernel<SORT_CAP_LARGE, true>), dim3(num_tiles), dim3(256), 0, st, ranges, values, splats,
                       list, scratch);
}

// ---------------------------------------------------------------------------------------------
// Bitmaps over the sorted list (one ballot per bitmap per 64 entries: bitmap q < 4 = "covers quad q", bitmap 4 =
// "covers any quad"), and -- after a prefix sum over the words' popcounts -- the compacted lists the blend
// kernels stream.
__global__ void __launch_bounds__(256)
bitmap_kernel(const uint64_t* __restrict__ list, int64_t N, uint64_t* __restrict__ bitmaps, uint32_t* __restrict__ wcount,
              size_t bitmap_words)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = i < N ? (uint32_t)list[i] : 0u;
    const size_t word = (size_t)(i >> 6);
#pragma unroll
    for (int q = 0; q < NUM_BITMAPS; ++q) {
        const uint64_t m = __ballot(q < 4 ? ((v >> (GID_BITS + q)) & 1u) : ((v >> GID_BITS) != 0u));
        if ((threadIdx.x & 63) == 0 && word < bitmap_words) {
            bitmaps[(size_t)q * bitmap_words + word] = m;
            wcount[(size_t)q * bitmap_words + word] = (uint32_t)__popcll(m);
        }
    }
}

// wprefix holds the INCLUSIVE scan of the word popcounts on entry and the exclusive one on exit.
__global__ void __launch_bounds__(256)
compact_kernel(const uint64_t* __restrict__ list, int64_t N, const uint64_t* __restrict__ bitmaps,
               uint32_t* __restrict__ wprefix, size_t bitmap_words, uint64_t* __restrict__ act)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const size_t word = (size_t)(i >> 6);
    if (word >= bitmap_words) return;  // whole wave
    const uint64_t entry = i < N ? list[i] : 0ull;  // (pos1 << 32) | quad mask << 28 | gaussian
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int q = 0; q < NUM_BITMAPS; ++q) {
        const size_t w = (size_t)q * bitmap_words + word;
        const uint64_t m = bitmaps[w];
        const uint32_t excl = wprefix[w] - (uint32_t)__popcll(m);
        if ((m >> lane) & 1ull) act[excl + (uint32_t)__popcll(m & lt)] = entry;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) wprefix[w] = excl;
    }
    // dead entries (gaussian 0, position 0) around the real ones: the blend kernels prefetch a few past either end
    if (i < ACT_PAD) {
        const uint32_t total = wprefix[NUM_BITMAPS * bitmap_words - 1];  // the last word is all zero: incl == excl
        act[total + i] = 0ull;
        act[(int64_t)i - ACT_PAD] = 0ull;
    }
}

void launch_bitmaps_and_compact(const uint64_t* list, int64_t N, uint64_t* bitmaps, size_t bitmap_words, uint32_t* wprefix,
                                uint32_t* scan_tmp, uint64_t* act, hipStream_t st)
{
    // the grid covers every bitmap word (also the zero words past N)
    const int64_t threads = (int64_t)bitmap_words * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(bitmap_kernel, dim3(blocks), dim3(256), 0, st, list, N, bitmaps, wprefix, bitmap_words);
    launch_scan_inclusive(wprefix, wprefix, scan_tmp, (int)(NUM_BITMAPS * bitmap_words), st);
    hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(256), 0, st, list, N, bitmaps, wprefix, bitmap_words, act);
}

}  // namespace hgs
