// The (Gaussian, tile) pair walk shared by the binning kernels (binning.hip) and the fused count at the tail of the
// preprocess kernel (preprocess.hip).
#pragma once
#include "hgs_common.h"

namespace hgs {

// All binning kernels walk the (Gaussian, tile) pairs the same way.  A workgroup owns BIN_GROUP Gaussians, one
// per thread, and a thread walks its own rectangle, record in registers (a typical splat touches 9-16 tiles, so the
// lanes of a wave finish together) -- no staging, no cross-lane search.  A splat with more than BIN_SOLO_MAX tiles is
// not walked by its lane: afterwards the wave takes such splats one at a time (record broadcast with v_readlane), 64
// tiles per step, so one huge splat cannot serialise a lane.  Atomics are aggregated per workgroup in an LDS array
// indexed by tile and touch global memory once per (workgroup, tile) with coalesced vector atomics: scattered global
// atomics cost ~15 G cache-line transactions/s on this chip, more than everything else in the binning phase together.
constexpr int BIN_THREADS = BIN_GROUP;    // threads of the fallback kernels (BIN_GROUP, BIN_LDS_TILES, bin_group_for: hgs_common.h)
constexpr uint32_t BIN_SOLO_MAX = 48;     // tiles a lane walks on its own

// The per-tile LDS counters of a binning group: 32-bit words or, on frames beyond BIN_LDS_TILES tiles (H16), 16-bit halves -- a group
// holds at most BIN_GROUP = 1 024 Gaussians and a Gaussian counts at most once per tile, so a half never carries into its neighbour.
template <bool H16>
struct TileHist {
    uint32_t* p;
    __device__ __forceinline__ void zero(int t) const
    {
        if constexpr (H16) reinterpret_cast<uint16_t*>(p)[t] = (uint16_t)0;
        else p[t] = 0u;
    }
    __device__ __forceinline__ uint32_t add(int t) const   // returns the count before
    {
        if constexpr (H16) {
            const uint32_t sh = ((uint32_t)t & 1u) << 4;
            return (atomicAdd(&p[t >> 1], 1u << sh) >> sh) & 0xFFFFu;
        } else
            return atomicAdd(&p[t], 1u);
    }
    __device__ __forceinline__ uint32_t get(int t) const
    {
        if constexpr (H16) return (uint32_t)reinterpret_cast<const uint16_t*>(p)[t];
        else return p[t];
    }
    static __host__ __device__ constexpr size_t bytes(int num_tiles) { return H16 ? 2u * (size_t)((num_tiles + 1) & ~1) : 4u * (size_t)num_tiles; }
};
static_assert(BIN_GROUP <= 0xFFFF, "16-bit per-tile counters of a binning group");

struct SplatRect {  // what the walk needs of one Gaussian
    float x, y, A, B, C, thr;  // centre, log2-domain half-conic, threshold on the exponent (emit only)
    uint32_t depth_bits;
    int minx, miny, width;
    uint32_t cnt;  // tiles touched (0: culled)
};

__device__ __forceinline__ SplatRect load_rect(int P, const Camera& cam, const Splat* __restrict__ splats, int g, bool with_mask_inputs)
{
    SplatRect r;
    r.x = r.y = 0.f, r.A = r.C = -1.f, r.B = 0.f, r.thr = 3.0e38f, r.depth_bits = 0, r.minx = r.miny = 0, r.width = 1, r.cnt = 0;
    if (g < P) {
        const float4 tail = reinterpret_cast<const float4*>(splats + g)[2];
        const int radius = __float_as_int(tail.z);
        if (radius > 0) {
            const float4 head = reinterpret_cast<const float4*>(splats + g)[0];
            const float px = head.x, py = head.y, radf = (float)radius;
            // identical expressions to the preprocess kernel => identical rectangle
            const int minx = (int)fminf((float)cam.gx, fmaxf(0.0f, (px - radf) / 16.0f));
            const int maxx = (int)fminf((float)cam.gx, fmaxf(0.0f, (px + radf + 15.0f) / 16.0f));
            const int miny = (int)fminf((float)cam.gy, fmaxf(0.0f, (py - radf) / 16.0f));
            const int maxy = (int)fminf((float)cam.gy, fmaxf(0.0f, (py + radf + 15.0f) / 16.0f));
            r.cnt = (uint32_t)((maxx - minx) * (maxy - miny));
            r.x = px, r.y = py;
            r.minx = minx, r.miny = miny, r.width = maxx - minx;
            r.depth_bits = __float_as_uint(tail.y);
            if (with_mask_inputs) {
                const float4 hc = reinterpret_cast<const float4*>(splats + g)[3];  // (ca, cb, cc, L): the half-conic quarter
                r.A = hc.x, r.B = hc.y, r.C = hc.z;
                // log2 domain (hgs_common.h): contributes iff exp2(power + L) >= 1/255  <=>  power >= -(log2 255 + L);
                // 0.07 of slack covers the blend kernels' rounding (and makes the mask a strict superset)
                r.thr = -(7.9943534f + hc.w) - 0.07f;
            }
        }
    }
    return r;
}

// f(owner_lane, tx, ty, rect of the owner) for every tile of every rectangle held by the wave's lanes
template <class F>
__device__ __forceinline__ void for_each_pair(const SplatRect& mine, F&& f)
{
    const int lane = threadIdx.x & 63;
    const bool big = mine.cnt > BIN_SOLO_MAX;
    if (!big && mine.cnt) {
        int tx = mine.minx, ty = mine.miny;
        const int endx = mine.minx + mine.width;
        for (uint32_t k = 0; k < mine.cnt; ++k) {
            f(lane, tx, ty, mine);
            if (++tx == endx) tx = mine.minx, ++ty;
        }
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(big);
    while (todo) {  // wave-uniform
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        SplatRect r;
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), src));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), src));
        r.A = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.A), src));
        r.B = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.B), src));
        r.C = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.C), src));
        r.thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.thr), src));
        r.depth_bits = (uint32_t)__builtin_amdgcn_readlane((int)mine.depth_bits, src);
        r.minx = __builtin_amdgcn_readlane(mine.minx, src);
        r.miny = __builtin_amdgcn_readlane(mine.miny, src);
        r.width = __builtin_amdgcn_readlane(mine.width, src);
        r.cnt = (uint32_t)__builtin_amdgcn_readlane((int)mine.cnt, src);
        for (uint32_t k = (uint32_t)lane; k < r.cnt; k += 64u) {
            const uint32_t ry = k / (uint32_t)r.width, rx = k - ry * (uint32_t)r.width;
            f(src, r.minx + (int)rx, r.miny + (int)ry, r);
        }
    }
}

}  // namespace hgs
