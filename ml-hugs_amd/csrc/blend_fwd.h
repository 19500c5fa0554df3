// Forward alpha blending of one 8x8 quad by one wave: shared by the stand-alone forward kernel
// (blend.hip) and the fused tile-sort + blend kernel (binning.hip).  See blend.hip for the design notes.
#pragma once
#include "hgs_common.h"

namespace hgs {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v4f* const_f4p;
typedef const __attribute__((address_space(4))) float* const_f32p;
typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
typedef const __attribute__((address_space(4))) uint64_t* const_u64p;
typedef const __attribute__((address_space(4))) v2u* const_u2p;

struct SplatRec {  // wave-uniform (lives in SGPRs); log2 domain: alpha = exp2(L - (A dx + B dy)^2 - (C dy)^2),
    float x, y, A, B, C, L, r, g, b;  // (A, B, C) = the conic's Cholesky factors (la, lb, lc) of hgs_common.h
};

// `entry_low` = low word of a list entry (mask << 28 | gaussian); the index is clamped because the software
// pipeline reads a few entries past either end of a list, where memory may hold anything
__device__ __forceinline__ SplatRec load_rec(const Splat* splats, uint32_t entry_low, uint32_t last_gaussian)
{
    const uint32_t gid = min(entry_low & GID_MASK, last_gaussian);
    // 32-bit byte offset (P * 64 < 2^32 is checked by the API): one shift + base+offset scalar loads, all in one cache line
    const_f4p p = (const_f4p)((const char*)splats + gid * 64u);
    const v4f h0 = p[0], h1 = p[1];
    const float b = ((const_f32p)p)[8];
    SplatRec s;
    s.x = h0.x, s.y = h0.y;
    s.A = h0.z, s.B = h0.w, s.C = h1.x;
    s.L = h1.y, s.r = h1.z, s.g = h1.w, s.b = b;
    return s;
}

// log2 of the uncapped alpha: L - (A dx + B dy)^2 - (C dy)^2, five VALU ops; never above L (hgs_common.h)
__device__ __forceinline__ float log2_alpha(const SplatRec& s, float dx, float dy)
{
    const float t = __builtin_fmaf(s.A, dx, s.B * dy);
    const float u = s.C * dy;
    return __builtin_fmaf(-t, t, __builtin_fmaf(-u, u, s.L));
}

// Tile order: workgroup / wave id == tile id.  Consecutive workgroups are dealt round-robin to the 8 XCDs, so every
// XCD gets every 8th tile of every image row: a little less L2 locality than one band of the image per XCD (+3 us in
// the forward on a uniform scene), but the XCDs stay evenly loaded when the Gaussians are not -- a person-sized blob in
// the middle of the frame cost the banded order 17 us (forward) / 19-49 us (backward).

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4u* const_u4p;

// two consecutive entries of a compacted list: {mask|gaussian 0, pos 0, mask|gaussian 1, pos 1} (one s_load_dwordx4)
__device__ __forceinline__ v4u load_pair(const uint64_t* act, uint32_t idx)
{
    return *(const_u4p)(act + idx);
}

// ------------------------------------------------------------------------------------------------
// One list entry applied to the wave's 64 pixels, fully predicated (v_cndmask, no exec-mask branches).
// T carries the "done" flag in its sign: a pixel that would fall below T_STOP keeps |T| and turns negative,
// after which test_T < 0 fails every later update.  `pos1` = 1-based position in the tile list (uniform).
__device__ __forceinline__ void fwd_accumulate(const SplatRec& s, uint32_t pos1, float pxf, float pyf, float& T,
                                               float& C0, float& C1, float& C2, uint32_t& last)
{
    const float dx = s.x - pxf, dy = s.y - pyf;
    const float e = log2_alpha(s, dx, dy);
    const float alpha = fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(e));
    const bool ok = alpha >= ALPHA_MIN;  // (the reference's `power > 0` test cannot fire: e <= L by construction)
    const float test_T = T * (1.0f - alpha);
    const bool upd = ok && test_T >= T_STOP;
    const float wgt = upd ? alpha * T : 0.0f;
    C0 = __builtin_fmaf(s.r, wgt, C0);
    C1 = __builtin_fmaf(s.g, wgt, C1);
    C2 = __builtin_fmaf(s.b, wgt, C2);
    T = upd ? test_T : (ok ? -__builtin_fabsf(T) : T);
    last = upd ? pos1 : last;
}

// First slot of tile `tile` (a list of `n_tile` entries) in the checkpoint buffer, or nullptr: the frame leaves no checkpoints,
// or it is a dense frame and this tile is not a deep one (hgs_common.h, CKPT_DEEP_MIN).  The workgroup also records which tile
// its slots belong to -- the backward's workgroups are dealt slots, not tiles (blend.hip) -- or that they hold nothing.
__device__ __forceinline__ float4* ckpt_begin(const Ckpt& ck, uint32_t tile, uint32_t n_tile)
{
    if (!ck.state) return nullptr;
    const bool leave = *(const_u32p)ck.sparse != 0u || n_tile >= CKPT_DEEP_MIN;
    const uint32_t first = ((const_u32p)ck.seg_first)[tile], end = ((const_u32p)ck.seg_first)[tile + 1];
    for (uint32_t k = first + threadIdx.x; k < end; k += blockDim.x) ck.slot_tile[k] = leave ? tile : CKPT_SLOT_NONE;
    return leave ? ck.state + (size_t)first * 256u : nullptr;
}

// One wave = the 64 pixels of quad `w` (0..3) of tile (tx, ty); walks `n` entries of the quad's compacted list.
// ck_tile != nullptr: the tile's checkpoint slots -- slot k receives the wave's state (T with its "done" sign, colour
// prefix) BEFORE list position (k + 1) * CKPT_SEG, the slot after the last full segment the state the wave ended with,
// and *nproc_out the number of entries it walked.
__device__ __forceinline__ void blend_forward_wave(const Camera& cam, uint32_t lastg, int tx, int ty, int w, uint32_t n,
                                                   const uint64_t* __restrict__ list, const Splat* __restrict__ splats,
                                                   const float* __restrict__ bg, float* __restrict__ out_color,
                                                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int clamp_output,
                                                   float4* __restrict__ ck_tile = nullptr, uint32_t* __restrict__ nproc_out = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;

    float T = inside ? 1.0f : -1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    uint32_t walked = 0;
    float4* ck_mine = ck_tile ? ck_tile + w * 64 + lane : nullptr;

    if (n) {
        // Software pipeline, two entries per half-iteration, two register sets (A/B) so nothing is copied:
        // while set A is blended, set B's records and the following pair of entries are in flight.
        v4u eA = load_pair(list, 0);
        SplatRec rA0 = load_rec(splats, eA.x, lastg), rA1 = load_rec(splats, eA.z, lastg);
        v4u eB = load_pair(list, 2);
        walked = n;
        for (uint32_t j = 0; j < n; j += 4) {
            if (ck_tile && j && (j & (uint32_t)(CKPT_SEG - 1)) == 0u) ck_mine[(size_t)((j >> CKPT_SHIFT) - 1u) * 256u] = make_float4(T, C0, C1, C2);
            const SplatRec rB0 = load_rec(splats, eB.x, lastg), rB1 = load_rec(splats, eB.z, lastg);
            const v4u eA2 = load_pair(list, j + 4);
            fwd_accumulate(rA0, eA.y, pxf, pyf, T, C0, C1, C2, last);
            if (j + 1 < n) fwd_accumulate(rA1, eA.w, pxf, pyf, T, C0, C1, C2, last);
            if (j + 2 >= n) break;
            if (__ballot(T > 0.0f) == 0ull) { walked = j + 2; break; }
            rA0 = load_rec(splats, eA2.x, lastg), rA1 = load_rec(splats, eA2.z, lastg);
            const v4u eB2 = load_pair(list, j + 6);
            fwd_accumulate(rB0, eB.y, pxf, pyf, T, C0, C1, C2, last);
            if (j + 3 < n) fwd_accumulate(rB1, eB.w, pxf, pyf, T, C0, C1, C2, last);
            if (__ballot(T > 0.0f) == 0ull) { walked = min(j + 4, n); break; }
            eA = eA2, eB = eB2;
        }
        // The next pair's records (rA0, rA1) are loaded in the MIDDLE of the loop body so that they fly while the second pair
        // is blended.  LLVM sinks loads whose results only the next trip uses below the "all pixels done" exit -- right in
        // front of the wait at the loop's top, where a wave that runs alone (the deep quads that outlast everybody on a
        // sparse frame) sits out the whole scalar-load latency once per trip.  Keeping the values alive past the loop (and
        // compiling binning.hip without the machine-sink pass, see the Makefile) pins the loads where they are written.
        asm volatile("" ::"s"(rA0.x), "s"(rA0.L), "s"(rA0.b), "s"(rA1.x), "s"(rA1.L), "s"(rA1.b));
    }
    if (ck_tile) {
        const uint32_t segs = (walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT;
        if (segs >= 2u) ck_mine[(size_t)(segs - 1u) * 256u] = make_float4(T, C0, C1, C2);
        if (lane == 0) *nproc_out = walked;
    }
    if (inside) {
        const size_t HW = (size_t)cam.H * cam.W, pix = (size_t)py * cam.W + px;
        const float Tf = __builtin_fabsf(T);
        final_T[pix] = Tf;
        float c0 = __builtin_fmaf(Tf, bg[0], C0), c1 = __builtin_fmaf(Tf, bg[1], C1), c2 = __builtin_fmaf(Tf, bg[2], C2);
        // bits 29..31 of n_contrib: "dL/dcolour passes" per channel -- all set without clamping, else set where the
        // unclamped value lies inside [0, 1] (torch.clamp's backward); last < 2^28 (GID_BITS)
        uint32_t pass = 7u;
        if (clamp_output) {
            pass = (c0 >= 0.0f && c0 <= 1.0f ? 1u : 0u) | (c1 >= 0.0f && c1 <= 1.0f ? 2u : 0u) | (c2 >= 0.0f && c2 <= 1.0f ? 4u : 0u);
            c0 = fminf(fmaxf(c0, 0.0f), 1.0f), c1 = fminf(fmaxf(c1, 0.0f), 1.0f), c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
        }
        n_contrib[pix] = last | (pass << 29);
        out_color[pix] = c0;
        out_color[HW + pix] = c1;
        out_color[2 * HW + pix] = c2;
    }
}

}  // namespace hgs
