// Per-tile alpha blending: forward (K6) and pixel-side backward (K7).  SURVEY.md A.4 / A.5.
//
// CDNA4 design (not the CUDA shape):
//  * No LDS staging, no barrier: the splat record of a list entry is the same for every lane, so it is fetched
//    through the scalar data cache (s_load_dwordx8 from the constant address space) straight into SGPRs and
//    used as a scalar operand of the VALU math.
//  * Coverage culling: the tile-sort kernel leaves, for every tile, five compacted lists -- the entries that can
//    reach alpha >= 1/255 in quad q (q = 0..3, the tile's four 8x8 quads) and those that reach any quad.  A wave
//    streams only its list, so the ~60 % of (quad, splat) pairs that cannot touch its pixels cost no instruction
//    at all (the tile list itself stays bit-identical to the reference's sorted list).
//  * The walk is software pipelined with two SGPR register sets: while one pair of entries is blended, the next
//    pair's records and the pair of entries after that are in flight (scalar loads return out of order, so the
//    single lgkmcnt(0) sits at the top of the half-iteration, before new loads are issued).
//  * log2 domain: the record holds log2e-scaled half-conics and log2(opacity), so alpha is one v_exp_f32 of a five-op
//    quadratic, with no multiply before or after the transcendental.
//  * Forward: a 16x16 tile is one 256-thread workgroup = 4 independent wave64s, one per quad; the per-pixel
//    update is fully predicated (v_cndmask), the "done" flag is the sign bit of T, early-out is per wave.
//  * Backward: ONE wave per tile; a lane owns four pixels, one in each quad, and the entry's quad mask selects
//    with scalar branches which of the four per-pixel evaluations run.  Per pixel only (T, S) are carried, S being
//    everything composited behind the entry dotted with dL/dpixel; per Gaussian nine RAW sums are accumulated (moments
//    of u = alpha_uncapped dL/dalpha, and the colour sums) -- the per-Gaussian factors are applied once per Gaussian by
//    preprocess_backward_kernel.  The nine partial sums of all covered quads are added in registers, so there is one
//    cross-lane reduction per (tile, entry) -- a transpose-reduce on gfx950's v_permlane32_swap / v_permlane16_swap
//    plus DPP adds that leaves the nine totals in nine lanes -- and ONE global_atomic_add_f32 instruction into a
//    contiguous [P][12] accumulator record.
//
// Compiled with -ffp-contract=off and -fno-slp-vectorize; the FMAs below are explicit so forward and backward
// evaluate alpha with the identical instruction sequence (backward must re-take forward's skip decisions).
#include <cstdlib>

#include "hgs_common.h"
#include "blend_fwd.h"

namespace hgs {

// Stand-alone forward (HGS_FUSED_SORT_BLEND=0, and the repair of a wrong "no long tiles" guess): the first `num_workers`
// workgroups blend the LONG tiles' quads split by depth (deep_forward_worker, blend_fwd.h: the same code path as in the fused
// sort + blend kernel of binning.hip, so a frame's image does not depend on which kernel blended it); behind them -- all_tiles --
// workgroup b = tile b, one wave per quad, for every tile that is not long.
__global__ void __launch_bounds__(256)
blend_forward_kernel(Camera cam, uint32_t lastg, const uint2* __restrict__ ranges, const uint64_t* __restrict__ act,
                     size_t act_stride, const uint32_t* __restrict__ act_count, const Splat* __restrict__ splats, const float* __restrict__ bg, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ n_total,
                     int clamp_output, const uint32_t* __restrict__ large_tiles, uint32_t num_workers, int long_sorted, Ckpt ck)
{
    __shared__ DeepShared deep;
    if (((const_u32p)n_total)[1]) return;  // binning buffer too small for this frame: the host re-runs it (hgs_api.hip)
    // (the scan's decision, binning.hip: 0 = no tile is blended split by depth, else the long tiles beyond that many entries are)
    const uint32_t deep_min = num_workers != 0u ? ((const_u32p)n_total)[8] : 0u;
    const bool deep_blend = deep_min != 0u;
    if (blockIdx.x < num_workers) {
        if (deep_blend)
            deep_forward_worker(blockIdx.x, num_workers, cam, lastg, ranges, act, act_stride, act_count, splats, bg, out_color, final_T,
                                n_contrib, clamp_output, ck, large_tiles, n_total, deep, gridDim.x == num_workers);
        // the repair pass (workers only in the grid): the long tiles that are NOT blended by depth -- all of them on a shallow sparse
        // frame, those of at most deep_min entries otherwise -- are taken one wave per quad, four waves = four quads of one tile each
        if (gridDim.x != num_workers) return;
        const uint32_t count = ((const_u32p)n_total)[2], threshold = ((const_u32p)n_total)[4];
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (uint32_t li = blockIdx.x; li < count; li += num_workers) {
            const uint32_t tile = ((const_u32p)large_tiles)[li];
            const v2u range = ((const_u2p)ranges)[tile];
            const uint32_t n_tile = range.y - range.x;
            if (n_tile <= threshold || (deep_blend && n_tile > deep_min)) continue;
            float4* ck_tile = ckpt_begin(ck, tile, n_tile);
            blend_forward_wave(cam, lastg, (int)(tile % (uint32_t)cam.gx), (int)(tile / (uint32_t)cam.gx), w,
                               ((const_u32p)act_count)[tile * NUM_LISTS + w], act + (size_t)w * act_stride + range.x, splats, bg, out_color,
                               final_T, n_contrib, clamp_output, ck_tile, ck.quad_nproc + tile * 4u + (uint32_t)w);
        }
        return;
    }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = (int)(blockIdx.x - num_workers);
    const v2u range = ((const_u2p)ranges)[tile];
    const uint32_t n_tile = range.y > range.x ? range.y - range.x : 0u;
    if (n_tile > ((const_u32p)n_total)[4]) {
        // a long tile: the workers have it -- or, when the long-tile sort was skipped on the caller's guess, nobody yet (its
        // pixels stay unwritten until the caller has repaired the guess)
        if (!long_sorted) return;
        if (deep_blend && n_tile > deep_min) {
            ckpt_begin(ck, (uint32_t)tile, n_tile);
            return;
        }
    }
    else if (long_sorted == 2) return;  // (only the long tiles: the repair pass without workers)
    const uint32_t n = n_tile ? ((const_u32p)act_count)[tile * NUM_LISTS + w] : 0u;
    float4* ck_tile = ckpt_begin(ck, (uint32_t)tile, n_tile);
    blend_forward_wave(cam, lastg, tile % cam.gx, tile / cam.gx, w, n, act + (size_t)w * act_stride + range.x, splats, bg, out_color,
                       final_T, n_contrib, clamp_output, ck_tile, ck.quad_nproc + (uint32_t)tile * 4u + (uint32_t)w);
}

// all_tiles: every tile (long ones through the workers when long_sorted); else only the long tiles (the repair)
void launch_blend_forward(const Camera& cam, int P, const uint2* ranges, const uint64_t* act, size_t act_stride,
                          const uint32_t* act_count, const Splat* splats, const float* bg, float* out_color,
                          float* final_T, uint32_t* n_contrib, const uint32_t* n_total, bool clamp_output,
                          const uint32_t* large_tiles, bool all_tiles, bool long_sorted, const Ckpt& ck, hipStream_t st)
{
    const int tiles = cam.gx * cam.gy;
    const bool deep = deep_forward_enabled();
    const uint32_t workers = long_sorted && deep ? deep_workers_for(tiles) : 0u;
    if (!all_tiles && !deep) {
        // the repair of a wrong "no long tiles" guess without the depth-parallel path: every tile's workgroup starts and those
        // of the tiles that are not long leave at once (the host does not know which tiles are long)
        hipLaunchKernelGGL(blend_forward_kernel, dim3(tiles), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges, act, act_stride, act_count,
                           splats, bg, out_color, final_T, n_contrib, n_total, clamp_output ? 1 : 0, large_tiles, 0u, 2, ck);
        return;
    }
    if (workers + (all_tiles ? tiles : 0) == 0) return;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(workers + (all_tiles ? tiles : 0)), dim3(256), 0, st, cam, (uint32_t)(P - 1),
                       ranges, act, act_stride, act_count, splats, bg, out_color, final_T, n_contrib, n_total, clamp_output ? 1 : 0,
                       large_tiles, workers, long_sorted ? 1 : 0, ck);
}

// ------------------------------------------------------------------------------------------------
// DPP helpers.  NB: every DPP move must be evaluated with all 64 lanes active and only then selected;
// inside a ?: arm the compiler would run it under a partial exec mask and read dead lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_ROR8 = 0x128;  // lane i <- lane i^8 (rotate by half a row)
constexpr int DPP_ROW_MIRROR = 0x140;       // lane i <- lane 15-i of its row
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <- lane 7-i of its half-row

// a and b are exchanged across the wave halves (W = 32) or across odd/even rows of 16 (W = 16) and added:
// W = 32: lanes 0-31 get a[l] + a[l+32], lanes 32-63 get b[l-32] + b[l];
// W = 16: rows 0 and 2 get a[row] + a[row+1], rows 1 and 3 get b[row-1] + b[row].
typedef uint32_t swap_pair_t __attribute__((ext_vector_type(2)));
template <int W>
__device__ __forceinline__ float swap_add(float a, float b)
{
    const uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    swap_pair_t r;
    if constexpr (W == 32) r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    else r = __builtin_amdgcn_permlane16_swap(ua, ub, false, false);
    const uint32_t r0 = r.x, r1 = r.y;
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

// pairwise transpose-reduce step: afterwards lanes with `hi` clear hold (a + partner's a) and lanes
// with `hi` set hold (b + partner's b); partner = lane ^ XOR within the quad.
template <int CTRL>
__device__ __forceinline__ float pair_step(float a, float b, bool hi)
{
    const float keep = hi ? b : a, send = hi ? a : b;
    return keep + dpp_mov<CTRL>(send);
}

// Per-pixel backward state of one lane; a lane owns FOUR pixels, one in each 8x8 quad of the tile
// (same (lx, ly) offset inside every quad), so that one wave covers the whole 16x16 tile.
struct PixBwd {
    float pxf, pyf;
    float T;           // running transmittance (in front of the entry being processed, once divided by 1 - alpha)
    float S;           // everything composited behind that entry, dotted with dL/dpixel:
                       //   T_final (bg . g) + sum over deeper entries j of (c_j . g) alpha_j T_j
    float g0, g1, g2;  // g = dL/dpixel
    uint32_t last_contributor;
};

// Adds pixel `p`'s contribution for splat `s` (list position pos1) to the lane-private partial sums v[9]; returns the
// lanes that took the splat (a scalar mask -- no per-lane "contributed" flag to carry).
// (dy, bdy = B dy, q = L - (C dy)^2: the row terms of log2_alpha, shared by the two quads of a tile row)
__device__ __forceinline__ unsigned long long bwd_pixel(const SplatRec& s, uint32_t pos1, PixBwd& p, float dy, float bdy,
                                                        float q, float (&v)[9])
{
    const float dx = s.x - p.pxf;
    const float t = __builtin_fmaf(s.A, dx, bdy);
    const float e = __builtin_fmaf(-t, t, q);  // == log2_alpha(s, dx, dy), same rounding; e <= L by construction
    const float alpha_uncapped = __builtin_amdgcn_exp2f(e);  // = opacity * G
    const float alpha = fminf(ALPHA_MAX, alpha_uncapped);
    // (ONE final compare -- of an alpha zeroed where the pixel has not been reached yet -- so that the ballot below is that
    //  compare's own SGPR result; the ballot of an AND of compares is materialised with a v_cndmask + v_cmp_ne pair)
    float alpha_if_reached = pos1 <= p.last_contributor ? alpha : 0.0f;
    asm("" : "+v"(alpha_if_reached));  // (opaque: otherwise the optimiser folds the select back into the AND)
    const bool act_lane = alpha_if_reached >= ALPHA_MIN;
    const unsigned long long took = __builtin_amdgcn_ballot_w64(act_lane);
    // (a branch-free body -- alphas forced to 0 for lanes that do not take the splat -- and a first-quad-assigns variant
    //  that spares the zero fill of v[] were both measured: equal or slower, the exec-masked region stays)
    if (act_lane) {
        const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
        p.T = p.T * inv;  // transmittance in front of this entry
        const float cg = __builtin_fmaf(s.r, p.g0, __builtin_fmaf(s.g, p.g1, s.b * p.g2));
        // pixel = sum_j c_j alpha_j T_j + T_final bg and every T_j behind this entry carries a factor (1 - alpha):
        //   dL/dalpha = T (c . g) - S / (1 - alpha)
        const float dL_dalpha = __builtin_fmaf(p.T, cg, -(p.S * inv));
        const float dch = alpha * p.T;
        p.S = __builtin_fmaf(cg, dch, p.S);
        // The geometry sums are kept in raw-moment form, u = opacity G dL/dalpha (straight-through alpha cap):
        //   v0 = sum u dx, v1 = sum u dy, v2 = sum u dx^2, v3 = sum u dx dy, v4 = sum u dy^2, v5 = sum u;
        // the per-Gaussian factors (opacity, conic, viewport scale, -1/2) are applied once per Gaussian by
        // preprocess_backward_kernel instead of once per pixel here.
        const float u = alpha_uncapped * dL_dalpha;
        const float ux = u * dx, uy = u * dy;
        v[0] += ux;
        v[1] += uy;
        v[2] = __builtin_fmaf(ux, dx, v[2]);
        v[3] = __builtin_fmaf(ux, dy, v[3]);
        v[4] = __builtin_fmaf(uy, dy, v[4]);
        v[5] += u;
        v[6] = __builtin_fmaf(dch, p.g0, v[6]);
        v[7] = __builtin_fmaf(dch, p.g1, v[7]);
        v[8] = __builtin_fmaf(dch, p.g2, v[8]);
    }
    return took;
}

// NQ = 4: one wave per tile (four tiles per 256-thread workgroup, no LDS, no barrier).  A lane's four pixels sit in
// the four quads, so the quad coverage mask of a list entry decides -- with scalar branches -- which of the
// four per-pixel evaluations run at all, while the nine partial sums of ALL covered quads are added up in
// registers before the single cross-lane reduction + atomic of that (tile, entry) pair.
// NQ = 1: one wave per QUAD (a tile is one workgroup), walking the quad's own compacted list: four times the waves,
// one reduction per (quad, entry) instead of per (tile, entry) -- more instructions in total, so it only pays when the
// frame has too few tiles to occupy the SIMDs (a 512x512 human-only render has 1 024 tiles for 1 024 SIMDs).
// SEG (with NQ = 1): the depth-segmented form for sparse frames -- the wave walks only entries [lo, hi) of its quad's
// list, starting from the forward blend's checkpoint behind entry hi - 1 (see blend_backward_segmented_kernel).
struct BwdSegment {
    uint32_t lo, hi;
    const float4* start;      // (T, colour prefix) of the wave's 64 pixels behind entry hi - 1; nullptr: the quad's last segment
    const float4* end_state;  // the state the forward wave ended with: its colour sums are the whole list's
};

template <int NQ, bool SEG = false>
__device__ __forceinline__ void
blend_backward_wave(const Camera& cam, uint32_t lastg, int tile, int w, v2u range, const uint64_t* __restrict__ act,
                    size_t act_stride, const uint32_t* __restrict__ act_count, const Splat* __restrict__ splats,
                    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                    const float* __restrict__ dL_dpix, float* __restrict__ grad_accum, const BwdSegment seg = BwdSegment{})
{
    static_assert(!SEG || NQ == 1, "segments are per quad");
    const int lane = threadIdx.x & 63;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int list_id = NQ == 4 ? 4 : w;  // the "any quad" list, or this wave's quad's
    const size_t HW = (size_t)cam.H * cam.W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    PixBwd p[NQ];
    uint32_t wmax = 0;
    // all twenty per-pixel loads are issued before any is consumed: out-of-image pixels read a clamped address and are
    // masked afterwards (a load under `if (inside)` costs one memory round trip per branch)
    float ld_T[NQ], ld_g[NQ][3];
    uint32_t ld_n[NQ];
    bool in_img[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int quad = NQ == 4 ? k : w;
        const int px = tx * TILE + (quad & 1) * 8 + (lane & 7);
        const int py = ty * TILE + (quad >> 1) * 8 + (lane >> 3);
        in_img[k] = px < cam.W && py < cam.H;
        const size_t pix = in_img[k] ? (size_t)py * cam.W + px : 0;
        p[k].pxf = (float)px, p[k].pyf = (float)py;
        ld_T[k] = final_T[pix], ld_n[k] = n_contrib[pix];
        ld_g[k][0] = dL_dpix[pix], ld_g[k][1] = dL_dpix[HW + pix], ld_g[k][2] = dL_dpix[2 * HW + pix];
    }
    float4 ck_start = make_float4(0.f, 0.f, 0.f, 0.f), ck_end = ck_start;
    if (SEG && seg.start) ck_start = seg.start[lane], ck_end = seg.end_state[lane];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const float Tf = in_img[k] ? ld_T[k] : 0.0f;
        p[k].T = Tf;
        // bits 29..31 of n_contrib: which channels' dL/dcolour pass (the forward's clamp mask)
        p[k].g0 = in_img[k] && (ld_n[k] >> 29 & 1u) ? ld_g[k][0] : 0.0f;
        p[k].g1 = in_img[k] && (ld_n[k] >> 30 & 1u) ? ld_g[k][1] : 0.0f;
        p[k].g2 = in_img[k] && (ld_n[k] >> 31 & 1u) ? ld_g[k][2] : 0.0f;
        p[k].S = Tf * (bg0 * p[k].g0 + bg1 * p[k].g1 + bg2 * p[k].g2);
        if (SEG && seg.start) {
            // behind the segment: the transmittance the forward had there (|.|: its sign is the forward's "done" flag, and a
            // pixel that was done takes none of this segment's entries) and everything composited behind it -- the whole
            // pixel's colour sums minus the prefix up to the checkpoint, dotted with dL/dpixel, plus the background term
            p[k].T = __builtin_fabsf(ck_start.x);
            p[k].S = __builtin_fmaf(p[k].g0, ck_end.y - ck_start.y,
                                    __builtin_fmaf(p[k].g1, ck_end.z - ck_start.z, __builtin_fmaf(p[k].g2, ck_end.w - ck_start.w, p[k].S)));
        }
        p[k].last_contributor = in_img[k] ? (ld_n[k] & 0x0FFFFFFFu) : 0u;
        wmax = max(wmax, p[k].last_contributor);
    }
    // the wave starts at the deepest entry any of its pixels composited (a per-QUAD cutoff on top of it was measured:
    // no gain on the dense joint render, +4 % on the uniform scene for the extra scalar compares)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    if (wmax == 0) return;
    const bool b3 = lane & 8;
    // which accumulator slot this lane's reduced value belongs to (see the reduction below); -1: none
    const int half_row = lane >> 3;  // row = half_row >> 1, half = half_row & 1
    const int slot_in_row[4] = {0, 2, 1, 3};
    const int slot_of_lane = lane == 63 ? 8 : ((lane & 7) == 0 ? slot_in_row[half_row >> 1] + 4 * (half_row & 1) : -1);

    // entries of this tile that cover at least one quad (list 4), walked back to front; those beyond the deepest
    // position any pixel composited (pos1 > wmax) are skipped with a scalar branch
    uint32_t n = SEG ? seg.hi - seg.lo : ((const_u32p)act_count)[tile * NUM_LISTS + list_id];
    if (n == 0) return;
    const uint64_t* first = act + (size_t)list_id * act_stride + range.x + (SEG ? seg.lo : 0u);
    if (SEG && (uint32_t)(((const_u64p)first)[0] >> 32) > wmax) return;  // the whole segment lies behind every pixel's last entry
    // The walk starts at the deepest entry any pixel composited, not at the end of the list: where the pixels saturate
    // early (a dense human blob: lists of ~700 entries, pixels done after ~250) most of the list lies beyond wmax, and
    // stepping over it entry by entry -- record fetch, compare, branch -- cost a fifth of the kernel.  The list is in
    // ascending position order: binary search (scalar loads) for the number of entries with pos1 <= wmax.
    if ((uint32_t)(((const_u64p)first)[n - 1] >> 32) > wmax) {
        uint32_t lo = 0, hi = n - 1;  // invariant: entries [0, lo) have pos1 <= wmax, entry hi has pos1 > wmax
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)(((const_u64p)first)[mid] >> 32) <= wmax) lo = mid + 1; else hi = mid;
        }
        n = lo;
        if (n == 0) return;
    }
    const uint64_t* top = first + n;  // one past the deepest entry to visit

    auto backward_entry = [&](const SplatRec& s, uint32_t val, uint32_t pos1) {
        if (pos1 > wmax) return;
        float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned long long took = 0ull;  // lanes (of any quad) that took the splat: scalar
        if constexpr (NQ == 4) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (((val >> (GID_BITS + 2 * r)) & 3u) == 0u) continue;
                const float dy = s.y - p[2 * r].pyf, bdy = s.B * dy, cdy = s.C * dy, q = __builtin_fmaf(-cdy, cdy, s.L);
                if ((val >> (GID_BITS + 2 * r)) & 1u) took |= bwd_pixel(s, pos1, p[2 * r], dy, bdy, q, v);
                if ((val >> (GID_BITS + 2 * r + 1)) & 1u) took |= bwd_pixel(s, pos1, p[2 * r + 1], dy, bdy, q, v);
            }
        } else {  // every entry of a quad's list covers the quad
            const float dy = s.y - p[0].pyf, bdy = s.B * dy, cdy = s.C * dy, q = __builtin_fmaf(-cdy, cdy, s.L);
            took = bwd_pixel(s, pos1, p[0], dy, bdy, q, v);
        }
        if (took == 0ull) return;
        // ---- transpose-reduce of v0..v7 over the wave: each step adds partner lanes AND halves the number of live
        // registers.  Lane-half and row exchanges are gfx950's v_permlane{32,16}_swap (no select needed: the swap
        // itself routes value a to one half and value b to the other), then one select step inside the row and three
        // plain DPP adds.  Afterwards every lane of half-row (row r, half h) holds the total of slot_of_lane.
        const float s0 = swap_add<32>(v[0], v[1]), s1 = swap_add<32>(v[2], v[3]);   // lanes 0-31: a, lanes 32-63: b
        const float s2 = swap_add<32>(v[4], v[5]), s3 = swap_add<32>(v[6], v[7]);
        const float t0 = swap_add<16>(s0, s1), t1 = swap_add<16>(s2, s3);           // rows: v0 v2 v1 v3 | v4 v6 v5 v7
        float y = pair_step<DPP_ROW_ROR8>(t0, t1, b3);                              // lanes 0-7 of a row: t0, 8-15: t1
        y += dpp_mov<DPP_ROW_HALF_MIRROR>(y);
        y += dpp_mov<DPP_QUAD_XOR2>(y);
        y += dpp_mov<DPP_QUAD_XOR1>(y);
        // v8 (blue): plain reduction; the classic gfx9 row broadcasts leave the wave total in row 3
        float vb = v[8] + dpp_mov<DPP_QUAD_XOR1>(v[8]);
        vb += dpp_mov<DPP_QUAD_XOR2>(vb);
        vb += dpp_mov<DPP_ROW_HALF_MIRROR>(vb);
        vb += dpp_mov<DPP_ROW_MIRROR>(vb);
        // rows 1,3 += lane 15 of the row before; rows 2,3 += lane 31.  Spelled out: a masked-row DPP add leaves the other
        // rows untouched in ONE instruction; through the builtin it becomes v_mov 0 + v_mov_dpp + v_add.
        asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(vb));
        asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(vb));
        // eight half-row leaders + lane 63 add the nine totals into the Gaussian's accumulator record with one
        // atomic instruction
        if (slot_of_lane >= 0) atomicAdd(grad_accum + (size_t)(val & GID_MASK) * 12u + slot_of_lane, lane == 63 ? vb : y);
    };

    // pair p = entries top[-2p-2] (shallower) and top[-2p-1] (deeper); the front pad of `act` makes the read
    // below `a` by the last, half-used pair harmless.  Two register sets (A/B): while one is consumed the
    // other's records and the next pair of entries are in flight.
    v4u eA = load_pair(top - 2, 0);
    SplatRec rA0 = load_rec(splats, eA.z, lastg), rA1 = load_rec(splats, eA.x, lastg);  // rA0: deeper, first
    v4u eB = load_pair(top - 4, 0);
    for (uint32_t j = 0; j < n; j += 4) {
        // Scalar loads return out of order, so the only wait there is is "all of them": it has to sit HERE, before the next
        // set's loads are issued -- the compiler puts it at the first use of set A, i.e. behind the loads of set B issued just
        // above that use, and a wave then sits out a whole scalar-cache round trip per half-iteration (hidden at eight waves
        // per SIMD, not on a sparse frame's deep quads).
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        const SplatRec rB0 = load_rec(splats, eB.z, lastg), rB1 = load_rec(splats, eB.x, lastg);
        const v4u eA2 = load_pair(top - 6 - j, 0);
        backward_entry(rA0, eA.z, eA.w);
        if (j + 1 < n) backward_entry(rA1, eA.x, eA.y);
        if (j + 2 >= n) break;
        rA0 = load_rec(splats, eA2.z, lastg), rA1 = load_rec(splats, eA2.x, lastg);
        const v4u eB2 = load_pair(top - 8 - j, 0);
        backward_entry(rB0, eB.z, eB.w);
        if (j + 3 < n) backward_entry(rB1, eB.x, eB.y);
        eA = eA2, eB = eB2;
    }
}

// NQ = 4: workgroup b takes tiles 4b .. 4b+3, one wave each.  NQ = 1: workgroup b takes tile b, one wave per quad -- for
// SPARSE frames, i.e. fewer non-empty tiles than would give each of the 1 024 SIMDs four waves (decided by
// tile_scan_kernel, travels to the host with N): a 512x512 human-only render, a person in front of an empty background.
template <int NQ>
__global__ void __launch_bounds__(256)
blend_backward_kernel(Camera cam, uint32_t lastg, const uint2* __restrict__ ranges, const uint64_t* __restrict__ act,
                      size_t act_stride, const uint32_t* __restrict__ act_count, const Splat* __restrict__ splats,
                      const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpix, float* __restrict__ grad_accum, uint32_t skip_from)
{
    const int num_tiles = cam.gx * cam.gy;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vbid = NQ == 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + w;
    if (vbid >= num_tiles) return;
    const int tile = vbid;
    const v2u range = ((const_u2p)ranges)[tile];
    if (range.y <= range.x) return;
    if (skip_from && range.y - range.x >= skip_from) return;  // a deep tile of a dense frame: the segmented kernel has it
    blend_backward_wave<NQ>(cam, lastg, tile, w, range, act, act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum);
}

// Depth-segmented backward for SPARSE frames (a 512x512 human-only render, a person in front of an empty background): few
// tiles, lists hundreds of entries deep, and blend_backward_kernel<1> lasts as long as its deepest quad's chain of
// dependent entries while most SIMDs idle.  The forward blend left a checkpoint every CKPT_SEG list positions (blend_fwd.h);
// here workgroup b takes checkpoint SLOT b = (tile, segment m), one wave per quad, and walks only entries
// [m CKPT_SEG, (m + 1) CKPT_SEG) of the quad's list from the state the forward had behind them -- the chain is at most
// CKPT_SEG entries long and the frame's (quad, entry) pairs spread over all SIMDs.  Same per-entry arithmetic, skip rules,
// reduction and atomics as the one-wave-per-quad kernel.
__device__ __forceinline__ void
blend_backward_slot(uint32_t slot, const Camera& cam, uint32_t lastg, const uint2* __restrict__ ranges, const uint64_t* __restrict__ act,
                    size_t act_stride, const Splat* __restrict__ splats, const float* __restrict__ bg, const float* __restrict__ final_T,
                    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ grad_accum, const Ckpt& ck)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (a dense frame's slots are packed, deep tiles only -- tile_scan_kernel --, and the grid may cover the layout's upper bound)
    if (slot >= ((const_u32p)ck.seg_first)[cam.gx * cam.gy]) return;
    const uint32_t tile = ((const_u32p)ck.slot_tile)[slot];
    if (tile == CKPT_SLOT_NONE) return;  // (a dense frame's tile that left no checkpoints: the one-wave-per-tile walk has it)
    const uint32_t first_slot = ((const_u32p)ck.seg_first)[tile];
    const uint32_t m = slot - first_slot;
    const uint32_t walked = ((const_u32p)ck.quad_nproc)[tile * 4u + (uint32_t)w];  // entries the forward wave walked
    const uint32_t segs = (walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT;
    if (m >= segs) return;
    const v2u range = ((const_u2p)ranges)[tile];
    BwdSegment seg;
    seg.lo = m << CKPT_SHIFT, seg.hi = min(walked, (m + 1u) << CKPT_SHIFT);
    seg.start = m + 1u < segs ? ck.state + (size_t)(first_slot + m) * 256u + w * 64 : nullptr;
    seg.end_state = ck.state + (size_t)(first_slot + segs - 1u) * 256u + w * 64;
    blend_backward_wave<1, true>(cam, lastg, (int)tile, w, range, act, act_stride, nullptr, splats, bg, final_T, n_contrib, dL_dpix,
                                 grad_accum, seg);
}

__global__ void __launch_bounds__(256)
blend_backward_segmented_kernel(Camera cam, uint32_t lastg, const uint2* __restrict__ ranges, const uint64_t* __restrict__ act,
                                size_t act_stride, const Splat* __restrict__ splats, const float* __restrict__ bg,
                                const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                const float* __restrict__ dL_dpix, float* __restrict__ grad_accum, Ckpt ck)
{
    // (the grid is the frame's slot count, (N >> CKPT_SHIFT) + tiles)
    blend_backward_slot(blockIdx.x, cam, lastg, ranges, act, act_stride, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, ck);
}

// A DENSE frame with deep tiles, both forms in ONE launch: the first dense_blocks workgroups are blend_backward_kernel<4>'s (four
// tiles each, deep tiles skipped), the rest blend_backward_segmented_kernel's (one checkpoint slot each).  As two launches on one
// stream the second waited for the first's last wave; both add into the same accumulator, so nothing orders them.  The long
// one-wave-per-tile walks start first and the short segment walks fill the SIMDs their tail leaves.
// (round 6) NO occupancy attribute: rounds 4-5 asked for amdgpu_waves_per_eu(8), which caps the kernel at 80 SGPRs -- the segment walk
// alone holds 77, the two forms together 98 -- and the compiler spilled 21 of them to VGPR lanes (46 without -fno-slp-vectorize), with
// 9 v_writelane and 51 v_readlane inside the walks' loops (profiles/r6p_mixed_kernel_resource_usage.txt).  A frame whose deep tiles are most of its work paid for it: a person in front of a
// 600 000-Gaussian scene at 1280x720 -- every tile beyond CKPT_DEEP_MIN, the whole backward in this kernel's segment half -- 404 us
// against 339 for the stand-alone segmented kernel on the same slots; without the attribute 360 (trained-scene profile 0.636 -> 0.617 ms,
// the step's joint render 0.681 -> 0.665; C4's joint render, no deep tiles, unchanged).  -DHGS_MIXED_ATTR=... for A/B builds.
#ifndef HGS_MIXED_ATTR
#define HGS_MIXED_ATTR
#endif
__global__ void __launch_bounds__(256) HGS_MIXED_ATTR
blend_backward_mixed_kernel(Camera cam, uint32_t lastg, const uint2* __restrict__ ranges, const uint64_t* __restrict__ act,
                            size_t act_stride, const uint32_t* __restrict__ act_count, const Splat* __restrict__ splats,
                            const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                            const float* __restrict__ dL_dpix, float* __restrict__ grad_accum, Ckpt ck, uint32_t dense_blocks)
{
    if (blockIdx.x >= dense_blocks) {
        // (a dense frame's slots are packed, deep tiles only -- tile_scan_kernel --; the grid is their number when the host still
        //  knew it, else the layout's upper bound)
        blend_backward_slot(blockIdx.x - dense_blocks, cam, lastg, ranges, act, act_stride, splats, bg, final_T, n_contrib, dL_dpix,
                            grad_accum, ck);
        return;
    }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = (int)blockIdx.x * 4 + w;
    if (tile >= cam.gx * cam.gy) return;
    const v2u range = ((const_u2p)ranges)[tile];
    if (range.y <= range.x || range.y - range.x >= CKPT_DEEP_MIN) return;
    blend_backward_wave<4>(cam, lastg, tile, w, range, act, act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum);
}

void launch_blend_backward(const Camera& cam, int P, const uint2* ranges, const uint64_t* act, size_t act_stride,
                           const uint32_t* act_count, bool sparse_frame, const Splat* splats, const float* bg,
                           const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, float* grad_accum,
                           const Ckpt& ck, int64_t num_rendered, int64_t dense_slots, hipStream_t st)
{
    const int num_tiles = cam.gx * cam.gy;
    const int force = switches().bwd_waves_per_tile;  // HGS_BWD_WAVES_PER_TILE = 1 / 4: measurement override
    const bool per_quad = force ? force == 4 : sparse_frame;
    if (ck.state) {
        // the forward left checkpoints: for every tile of a sparse frame, for the deep tiles (CKPT_DEEP_MIN) of a dense one --
        // whose other tiles go through the one-wave-per-tile kernel as always (both add into the same accumulator)
        uint32_t slots = (uint32_t)(num_rendered >> CKPT_SHIFT) + (uint32_t)num_tiles;   // (the sparse layout; a dense frame's upper bound)
        if (!sparse_frame && dense_slots >= 0 && (uint64_t)dense_slots < slots) slots = (uint32_t)dense_slots;
        const bool two_launches = switches().bwd_two_launches;   // (A/B measurements and the equivalence test)
        if ((sparse_frame || two_launches) && slots)
            hipLaunchKernelGGL(blend_backward_segmented_kernel, dim3(slots), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges, act,
                               act_stride, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, ck);
        if (!sparse_frame && two_launches)
            hipLaunchKernelGGL(blend_backward_kernel<4>, dim3((num_tiles + 3) / 4), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges,
                               act, act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, CKPT_DEEP_MIN);
        else if (!sparse_frame) {
            const uint32_t dense_blocks = (uint32_t)(num_tiles + 3) / 4u;
            hipLaunchKernelGGL(blend_backward_mixed_kernel, dim3(dense_blocks + slots), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges,
                               act, act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, ck, dense_blocks);
        }
    } else if (per_quad)
        hipLaunchKernelGGL(blend_backward_kernel<1>, dim3(num_tiles), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges, act,
                           act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, 0u);
    else
        hipLaunchKernelGGL(blend_backward_kernel<4>, dim3((num_tiles + 3) / 4), dim3(256), 0, st, cam, (uint32_t)(P - 1), ranges,
                           act, act_stride, act_count, splats, bg, final_T, n_contrib, dL_dpix, grad_accum, 0u);
}

}  // namespace hgs
