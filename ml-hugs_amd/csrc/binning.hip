// Tile binning: prefix scan (K2), key emission (K3), stable LSD radix sort (K4), tile ranges (K5).
// Algorithm: SURVEY.md Appendix A.4.  Everything here is integer work and is compared bit-for-bit
// with the CPU oracle.  Wave64 ballots / popcounts do the per-digit ranking; no CUB/rocPRIM.
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
// K3: key emission, wave-balanced, plus a per-entry quad coverage mask.
//
// A wave owns 64 consecutive Gaussians; their tile counts are prefix-summed in registers and the wave's
// output slots are dealt to lanes 64 at a time (each slot finds its Gaussian by a 6-step search over the
// wave's prefix sums), so a single huge splat no longer serialises one lane and writes are coalesced.
// Order inside a Gaussian is y outer, x inner (A.4), so (key,value) slots are exactly the oracle's.
//
// The value's top 4 bits carry a coverage mask: bit q is set when the splat can reach alpha >= 1/255 on
// some pixel of the tile's 8x8 quad q (q = qx + 2 qy).  It is CONSERVATIVE (may be set needlessly, never
// missing): the blend kernels skip a (wave, splat) pair whose bit is clear without touching a VGPR.
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
emit_keys_kernel(int P, Camera cam, const Splat* __restrict__ splats, const uint32_t* __restrict__ order,
                 const uint32_t* __restrict__ offsets, uint32_t* __restrict__ keys, uint32_t* __restrict__ values)
{
    __shared__ float4 stage[4][64][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + w) * 64;  // first depth RANK of this wave
    if (g0 >= P) return;                       // whole wave
    const int g = g0 + lane < P ? (int)order[g0 + lane] : P;  // the Gaussian holding rank g0 + lane

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
            r1 = make_float4(mid.x, thr, __int_as_float(g), 0.f);
            r2 = make_float4(__int_as_float(minx), __int_as_float(miny), __int_as_float(maxx - minx), 0.f);
        }
    }
    const uint32_t incl = wave_inclusive_scan(cnt);
    r1.w = __uint_as_float(incl - cnt);
    stage[w][lane][0] = r0, stage[w][lane][1] = r1, stage[w][lane][2] = r2;
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    const uint32_t wave_base = g0 == 0 ? 0u : offsets[g0 - 1];

    for (uint32_t s0 = 0; s0 < total; s0 += 64) {
        const uint32_t s = s0 + lane;
        // owner = smallest lane whose inclusive count exceeds s (all lanes take part in the shuffles)
        int lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t v = (uint32_t)__shfl((int)incl, lo + step - 1, 64);
            if (v <= s) lo += step;
        }
        if (s < total) {
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
            keys[wave_base + s] = (uint32_t)(ty * cam.gx + tx);
            values[wave_base + s] = (mask << GID_BITS) | (uint32_t)__float_as_int(b.z);
        }
    }
}

void launch_emit_keys(int P, const Camera& cam, const Splat* splats, const uint32_t* order, const uint32_t* offsets,
                      uint32_t* keys, uint32_t* values, hipStream_t st)
{
    hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, cam, splats, order, offsets, keys,
                       values);
}

// ---------------------------------------------------------------------------------------------
// K4: LSD radix sort, digits of <= 9 bits, three kernels per pass:
//   upsweep   per-block digit histogram -> hist[digit][block], global digit totals
//   scan      one workgroup per digit: exclusive scan of its row + base of all smaller digits
//   downsweep stable scatter: rank = global base + earlier waves + earlier rounds + lower lanes
// Block = 256 threads = 4 waves; wave w owns a contiguous run of 16 rounds x 64 keys, so the
// (wave, round, lane) order is the input order and the sort is stable.

struct SortPlan {
    int passes;
    int bits[8];
    int shift[8];
};
static SortPlan make_plan(int num_bits)
{
    SortPlan p;
    if (num_bits < 1) num_bits = 1;
    p.passes = (num_bits + 8) / 9;
    int base = num_bits / p.passes, extra = num_bits % p.passes, sh = 0;
    for (int i = 0; i < p.passes; ++i) {
        p.bits[i] = base + (i < extra ? 1 : 0);
        p.shift[i] = sh;
        sh += p.bits[i];
    }
    return p;
}
int sort_input_buffer(int num_bits) { return make_plan(num_bits).passes & 1; }

template <typename KeyT>
__global__ void __launch_bounds__(SORT_THREADS)
sort_upsweep_kernel(const KeyT* __restrict__ keys, int64_t N, int shift, int nbits, uint32_t* __restrict__ hist,
                    uint32_t* __restrict__ totals, int nblocks)
{
    __shared__ uint32_t h[SORT_MAX_BINS];
    const int nbins = 1 << nbits;
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    const uint32_t mask = (uint32_t)nbins - 1u;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        int64_t i = base + (int64_t)r * SORT_THREADS + threadIdx.x;
        if (i < N) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) {
        uint32_t c = h[d];
        hist[(size_t)d * nblocks + blockIdx.x] = c;
        if (c) atomicAdd(&totals[d], c);
    }
}

__global__ void __launch_bounds__(256)
sort_scan_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, int nblocks)
{
    __shared__ uint32_t wsum[4];
    const int d = blockIdx.x;
    // base = sum of totals of smaller digits
    uint32_t part = 0;
    for (int k = threadIdx.x; k < d; k += 256) part += totals[k];
    uint32_t base;
    block_inclusive_scan(part, wsum, base);
    uint32_t* row = hist + (size_t)d * nblocks;
    uint32_t carry = base;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        int b = b0 + threadIdx.x;
        uint32_t v = b < nblocks ? row[b] : 0;
        uint32_t total;
        uint32_t inc = block_inclusive_scan(v, wsum, total);
        if (b < nblocks) row[b] = carry + inc - v;
        carry += total;
    }
}

// vals_in == nullptr: the value of element i is i itself (first pass of an argsort)
template <typename KeyT>
__global__ void __launch_bounds__(SORT_THREADS)
sort_downsweep_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                      KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t N, int shift,
                      int nbits, const uint32_t* __restrict__ hist, int nblocks)
{
    __shared__ uint32_t cnt[4][SORT_MAX_BINS];
    __shared__ uint32_t gbase[SORT_MAX_BINS];
    const int nbins = 1 << nbits;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < 4 * SORT_MAX_BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) gbase[d] = hist[(size_t)d * nblocks + blockIdx.x];
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * (SORT_ITEMS * 64);
    const uint32_t mask = (uint32_t)nbins - 1u;
    const uint64_t lt = (1ull << lane) - 1ull;
    KeyT key[SORT_ITEMS];
    uint32_t loc[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < N;
        key[r] = valid ? keys_in[i] : (KeyT)~(KeyT)0;
        const uint32_t digit = (uint32_t)(key[r] >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt);
        uint32_t pre = 0;
        if (valid) pre = cnt[w][digit];
        // in-order LDS within the wave: every peer has read `pre` before the leader's update lands
        if (valid && before == 0) cnt[w][digit] = pre + (uint32_t)__popcll(peers);
        loc[r] = pre + before;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nbins; d += SORT_THREADS) {
        uint32_t run = gbase[d];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t c = cnt[k][d];
            cnt[k][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        if (i < N) {
            const uint32_t digit = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = cnt[w][digit] + loc[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in ? vals_in[i] : (uint32_t)i;
        }
    }
}

template <typename KeyT>
static void sort_pairs_impl(KeyT* keys_a, KeyT* keys_b, uint32_t* vals_a, uint32_t* vals_b, bool iota_values,
                            uint32_t* hist, uint32_t* totals, int64_t N, int num_bits, hipStream_t st)
{
    if (N <= 0) return;
    SortPlan p = make_plan(num_bits);
    const int nblocks = (int)((N + SORT_TILE - 1) / SORT_TILE);
    (void)hipMemsetAsync(totals, 0, sizeof(uint32_t) * SORT_MAX_BINS * 8, st);
    KeyT *kin = (p.passes & 1) ? keys_b : keys_a, *kout = (p.passes & 1) ? keys_a : keys_b;
    uint32_t *vin = (p.passes & 1) ? vals_b : vals_a, *vout = (p.passes & 1) ? vals_a : vals_b;
    for (int i = 0; i < p.passes; ++i) {
        uint32_t* tot = totals + (size_t)i * SORT_MAX_BINS;
        hipLaunchKernelGGL(sort_upsweep_kernel<KeyT>, dim3(nblocks), dim3(SORT_THREADS), 0, st, kin, N, p.shift[i],
                           p.bits[i], hist, tot, nblocks);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1 << p.bits[i]), dim3(256), 0, st, hist, tot, nblocks);
        hipLaunchKernelGGL(sort_downsweep_kernel<KeyT>, dim3(nblocks), dim3(SORT_THREADS), 0, st, kin,
                           (i == 0 && iota_values) ? (const uint32_t*)nullptr : vin, kout, vout, N, p.shift[i], p.bits[i],
                           hist, nblocks);
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
}

void launch_sort_pairs32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, bool iota_values,
                         uint32_t* hist, uint32_t* totals, int64_t N, int num_bits, hipStream_t st)
{
    sort_pairs_impl<uint32_t>(keys_a, keys_b, vals_a, vals_b, iota_values, hist, totals, N, num_bits, st);
}

// tt_sorted[r] = tiles_touched[order[r]]  (tile counts in depth order, input of the offsets scan)
__global__ void __launch_bounds__(256)
gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ order, uint32_t* __restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[order[i]];
}
void launch_gather_u32(const uint32_t* src, const uint32_t* order, uint32_t* dst, int n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(gather_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, order, dst, n);
}

// ---------------------------------------------------------------------------------------------
// K5: tile ranges, the four per-quad bitmaps over the sorted list (one ballot per quad per 64 entries), and --
// after a prefix sum over the bitmap words' popcounts -- the compacted per-quad lists the blend kernels stream.
__global__ void __launch_bounds__(256)
tile_ranges_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ values, int64_t N,
                   uint2* __restrict__ ranges, uint64_t* __restrict__ bitmaps, uint32_t* __restrict__ wcount,
                   size_t bitmap_words)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = 0;
    if (i < N) {
        const uint32_t t = keys[i];
        if (i == 0 || keys[i - 1] != t) ranges[t].x = (uint32_t)i;
        if (i == N - 1 || keys[i + 1] != t) ranges[t].y = (uint32_t)(i + 1);
        v = values[i];
    }
    const size_t word = (size_t)(i >> 6);
#pragma unroll
    for (int q = 0; q < NUM_BITMAPS; ++q) {  // q < 4: quad q covered; q == 4: any quad covered
        const uint64_t m = __ballot(q < 4 ? ((v >> (GID_BITS + q)) & 1u) : ((v >> GID_BITS) != 0u));
        if ((threadIdx.x & 63) == 0 && word < bitmap_words) {
            bitmaps[(size_t)q * bitmap_words + word] = m;
            wcount[(size_t)q * bitmap_words + word] = (uint32_t)__popcll(m);
        }
    }
}

// wprefix holds the INCLUSIVE scan of the word popcounts on entry and the exclusive one on exit.
__global__ void __launch_bounds__(256)
compact_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ values, int64_t N,
               const uint2* __restrict__ ranges, const uint64_t* __restrict__ bitmaps, uint32_t* __restrict__ wprefix,
               size_t bitmap_words, uint64_t* __restrict__ act)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const size_t word = (size_t)(i >> 6);
    if (word >= bitmap_words) return;  // whole wave
    uint64_t entry = 0;  // (pos1 << 32) | quad mask << 28 | gaussian
    if (i < N) {
        const uint32_t t = keys[i];
        const uint32_t pos1 = (uint32_t)i - ranges[t].x + 1u;
        entry = ((uint64_t)pos1 << 32) | (uint64_t)values[i];
    }
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
    // dead entries (gaussian 0, position 0) behind the last real one: the blend kernels prefetch a few past the end
    if (i == 0) {
        const uint32_t total = wprefix[NUM_BITMAPS * bitmap_words - 1];  // the last word is all zero: incl == excl
        for (int k = 0; k < ACT_PAD; ++k) act[total + k] = 0ull;
    }
}

void launch_tile_ranges(const uint32_t* keys, const uint32_t* values, int64_t N, uint2* ranges, int num_tiles,
                        uint64_t* bitmaps, size_t bitmap_words, uint32_t* wprefix, uint32_t* scan_tmp, uint64_t* act,
                        hipStream_t st)
{
    (void)hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, st);
    (void)hipMemsetAsync(act - ACT_PAD, 0, sizeof(uint64_t) * ACT_PAD, st);
    // the grid covers every bitmap word (also the zero words past N)
    const int64_t threads = (int64_t)bitmap_words * 64;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks), dim3(256), 0, st, keys, values, N, ranges, bitmaps, wprefix,
                       bitmap_words);
    launch_scan_inclusive(wprefix, wprefix, scan_tmp, (int)(NUM_BITMAPS * bitmap_words), st);
    hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(256), 0, st, keys, values, N, ranges, bitmaps, wprefix,
                       bitmap_words, act);
}

}  // namespace hgs
