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
template <typename Rec>
__device__ __forceinline__ void fwd_accumulate(const Rec& s, uint32_t pos1, float pxf, float pyf, float& T,
                                               float& C0, float& C1, float& C2, uint32_t& last)
{
    const float dx = s.x - pxf, dy = s.y - pyf;
    const float tq = __builtin_fmaf(s.A, dx, s.B * dy), uq = s.C * dy;
    const float e = __builtin_fmaf(-tq, tq, __builtin_fmaf(-uq, uq, s.L));   // == log2_alpha(s, dx, dy)
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

// Final state of one pixel -> image, final_T, n_contrib (T carries the "done" flag in its sign).
__device__ __forceinline__ void write_pixel(const Camera& cam, int px, int py, float T, float C0, float C1, float C2, uint32_t last,
                                            const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ final_T,
                                            uint32_t* __restrict__ n_contrib, int clamp_output)
{
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

// Which tiles leave checkpoints is the scan's word (n_total[3], binning.hip): 1 = every tile (a sparse frame), 0 = the deep tiles of a
// dense frame (hgs_common.h, CKPT_DEEP_MIN), 2 = none (a sparse frame whose backward runs one wave per quad from the end of the lists:
// CKPT_KIND_NONE)
__device__ __forceinline__ bool ckpt_leave(const Ckpt& ck, uint32_t n_tile)
{
    const uint32_t kind = *(const_u32p)ck.sparse;
    return kind == CKPT_KIND_ALL || (kind == CKPT_KIND_DEEP && n_tile >= CKPT_DEEP_MIN);
}

// First slot of tile `tile` (a list of `n_tile` entries) in the checkpoint buffer, or nullptr: the frame leaves no checkpoints,
// or it is a dense frame and this tile is not a deep one (hgs_common.h, CKPT_DEEP_MIN).  The workgroup also records which tile
// its slots belong to -- the backward's workgroups are dealt slots, not tiles (blend.hip) -- or that they hold nothing.
__device__ __forceinline__ float4* ckpt_begin(const Ckpt& ck, uint32_t tile, uint32_t n_tile)
{
    if (!ck.state || *(const_u32p)ck.sparse == CKPT_KIND_NONE) return nullptr;   // (no checkpoints: the slot layout is not even valid)
    const bool leave = ckpt_leave(ck, n_tile);
    const uint32_t first = ((const_u32p)ck.seg_first)[tile], end = ((const_u32p)ck.seg_first)[tile + 1];
    for (uint32_t k = first + threadIdx.x; k < end; k += blockDim.x) ck.slot_tile[k] = leave ? tile : CKPT_SLOT_NONE;
    return leave ? ck.state + (size_t)first * 256u : nullptr;
}

// What a wave's walk over its quad's list leaves per pixel (T with its "done" sign), and how far the wave walked.
struct FwdWalk { float T, C0, C1, C2; uint32_t last, walked; };

// The tail of a quad's blend: end-state checkpoint, walked count, pixel outputs.
__device__ __forceinline__ void blend_forward_finish(const FwdWalk& r, int W, int H, int px, int py, int lane, float4* __restrict__ ck_mine,
                                                     uint32_t* __restrict__ nproc_out, const float* __restrict__ bg,
                                                     float* __restrict__ out_color, float* __restrict__ final_T,
                                                     uint32_t* __restrict__ n_contrib, int clamp_output)
{
    if (ck_mine) {
        const uint32_t segs = (r.walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT;
        if (segs >= 2u) ck_mine[(size_t)(segs - 1u) * 256u] = make_float4(r.T, r.C0, r.C1, r.C2);
        if (lane == 0) *nproc_out = r.walked;
    }
    if (px < W && py < H) {
        Camera c;
        c.W = W, c.H = H;
        write_pixel(c, px, py, r.T, r.C0, r.C1, r.C2, r.last, bg, out_color, final_T, n_contrib, clamp_output);
    }
}

// One wave = the 64 pixels of a quad (pixel (pxf, pyf) per lane; `inside`: the pixel exists); walks `n` entries of the quad's
// compacted list, records fetched through the scalar cache.  ck_mine != nullptr: the lane's place in the tile's checkpoint slots --
// slot k receives the wave's state (T with its "done" sign, colour prefix) BEFORE list position (k + 1) * CKPT_SEG.
__device__ __forceinline__ FwdWalk blend_forward_walk(uint32_t lastg, float pxf, float pyf, bool inside, uint32_t n,
                                                      const uint64_t* __restrict__ list, const Splat* __restrict__ splats,
                                                      float4* __restrict__ ck_mine)
{
    float T = inside ? 1.0f : -1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    uint32_t walked = 0;
    if (n) {
        // Software pipeline, two entries per half-iteration, two register sets (A/B) so nothing is copied:
        // while set A is blended, set B's records and the following pair of entries are in flight.
        v4u eA = load_pair(list, 0);
        SplatRec rA0 = load_rec(splats, eA.x, lastg), rA1 = load_rec(splats, eA.z, lastg);
        v4u eB = load_pair(list, 2);
        walked = n;
        for (uint32_t j = 0; j < n; j += 4) {
            if (ck_mine && j && (j & (uint32_t)(CKPT_SEG - 1)) == 0u) ck_mine[(size_t)((j >> CKPT_SHIFT) - 1u) * 256u] = make_float4(T, C0, C1, C2);
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
    return FwdWalk{T, C0, C1, C2, last, walked};
}

// walk + finish for quad `w` (0..3) of tile (tx, ty)
__device__ __forceinline__ void blend_forward_wave(const Camera& cam, uint32_t lastg, int tx, int ty, int w, uint32_t n,
                                                   const uint64_t* __restrict__ list, const Splat* __restrict__ splats,
                                                   const float* __restrict__ bg, float* __restrict__ out_color,
                                                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int clamp_output,
                                                   float4* __restrict__ ck_tile = nullptr, uint32_t* __restrict__ nproc_out = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    float4* ck_mine = ck_tile ? ck_tile + w * 64 + lane : nullptr;
    const FwdWalk r = blend_forward_walk(lastg, (float)px, (float)py, px < cam.W && py < cam.H, n, list, splats, ck_mine);
    blend_forward_finish(r, cam.W, cam.H, px, py, lane, ck_mine, nproc_out, bg, out_color, final_T, n_contrib, clamp_output);
}

// ------------------------------------------------------------------------------------------------
// DEPTH-PARALLEL forward blend for LONG tiles (round 4).  A lone wave walking a deep quad's list is a chain of dependent
// entries: ~430 entries x 22 instructions at a lone wave's issue rate set the length of the whole kernel on the HUGS human
// renders (hugs/renderer/gs_renderer.py:56-82), with most SIMDs idle.  Alpha compositing is associative, so here a workgroup
// of DEEP_LANES waves takes ONE quad of a long tile and splits its list by depth, CKPT_SEG entries per wave and round:
//   phase A  wave m composes its entries from T = 1: local transmittance product P_m, local colour C_m, local last
//            contributor (the alpha < 1/255 skip is local; the T < 1e-4 stop is not, and is left out here) -> LDS
//   barrier
//   phase B  every wave forms, per pixel, T_start(j) = T_start(j - 1) P_(j-1) over the round's segments and finds the segment
//            m* in which the pixel stops (first j with T_start(j) P_j < 1e-4); segments in front of m* are added whole,
//            T_start(j) C_j; wave m* RE-WALKS its entries with the exact stop rule from T_start(m*) for the pixels that stop
//            there and publishes their final state; the per-pixel checkpoints the depth-segmented backward reads (blend.hip)
//            are the states of this scan.
// One barrier per round (double-buffered exchange), no launches per round, wasted work bounded by one round; the round loop
// ends when every pixel has stopped or the list is exhausted.  Which path a quad takes depends on its tile's list length only
// (tile_scan_kernel's threshold), so frames stay deterministic.  Against the one-wave walk the products are rounded in another
// order (~1e-7 relative); a pixel whose T lands within that of 1e-4 may stop one contributing entry earlier or later.
// -DHGS_TRACE (A/B builds only, tools/trace_fused.py): per-workgroup wall-clock stamps of the fused kernel
#if defined(HGS_TRACE) && defined(HGS_TRACE_THIS_FILE)   // (binning.hip defines the buffer pointer in front of this header)
#define HGS_TRACE_PUT(slot, val) do { if (g_trace_buf && threadIdx.x == 0) g_trace_buf[(size_t)blockIdx.x * 8u + (slot)] = (unsigned long long)(val); } while (0)
#define HGS_TRACE_ADD(slot, val) do { if (g_trace_buf && threadIdx.x == 0) g_trace_buf[(size_t)blockIdx.x * 8u + (slot)] += (unsigned long long)(val); } while (0)
#else
#define HGS_TRACE_PUT(slot, val) do { } while (0)
#define HGS_TRACE_ADD(slot, val) do { } while (0)
#endif
#ifndef DEEP_INLINE
#define DEEP_INLINE __forceinline__
#endif
constexpr int DEEP_LANES = 4;
constexpr uint32_t DEEP_ROUND = (uint32_t)DEEP_LANES * (uint32_t)CKPT_SEG;
// A wave's CKPT_SEG entries of the round are STAGED in LDS: lane l fetches entry l and its splat record with vector loads --
// all of the segment's records in flight at once, and the next round's already under way while this one is composed -- instead
// of one scalar-cache round trip per pair of entries (what a lone wave's walk spends most of its time waiting for: measured
// 145-190 ns per entry against ~90 ns of instructions); phase A and the re-walk of phase B both read the staged records back
// as broadcast LDS reads into VGPRs.
struct StagedRec { float x, y, A, B, C, L, r, g, b; uint32_t pos1; uint32_t pad0, pad1; };   // 48 bytes: three 16-byte reads
static_assert(sizeof(StagedRec) == 48, "StagedRec layout");
struct DeepShared {
    float4 comp[2][DEEP_LANES][64];        // (P_m, C_m) of the round, double-buffered
    uint32_t comp_last[2][DEEP_LANES][64]; // local last contributor (1-based tile position; 0: none)
    float4 fin[64];                        // final (T with its "done" sign, colour) of the pixels that stopped
    uint32_t fin_last[64];
    StagedRec stage[DEEP_LANES][CKPT_SEG];
};
constexpr size_t DEEP_LDS_BYTES = 17664;
static_assert(sizeof(DeepShared) == DEEP_LDS_BYTES, "the fused kernel's LDS buffer is sized for it");

// one list entry applied to the wave's 64 pixels: STOP = the exact rule (fwd_accumulate), else the compose step -- the same
// without the stop rule (T never falls "done"; P may underflow to 0, which only ever means "stops here")
template <bool STOP, typename Rec>
__device__ __forceinline__ void deep_accumulate(const Rec& s, uint32_t pos1, float pxf, float pyf, float& T, float& C0, float& C1,
                                                float& C2, uint32_t& last)
{
    const float dx = s.x - pxf, dy = s.y - pyf;
    const float t = __builtin_fmaf(s.A, dx, s.B * dy);
    const float u = s.C * dy;
    const float e = __builtin_fmaf(-t, t, __builtin_fmaf(-u, u, s.L));   // == log2_alpha
    const float alpha = fminf(ALPHA_MAX, __builtin_amdgcn_exp2f(e));
    const bool ok = alpha >= ALPHA_MIN;
    const float test_T = T * (1.0f - alpha);
    const bool upd = STOP ? (ok && test_T >= T_STOP) : ok;
    const float wgt = upd ? alpha * T : 0.0f;
    C0 = __builtin_fmaf(s.r, wgt, C0);
    C1 = __builtin_fmaf(s.g, wgt, C1);
    C2 = __builtin_fmaf(s.b, wgt, C2);
    if constexpr (STOP) T = upd ? test_T : (ok ? -__builtin_fabsf(T) : T);
    else T = upd ? test_T : T;
    last = upd ? pos1 : last;
}

// the wave's `n` staged entries (broadcast reads: every lane reads the same record) applied in order
template <bool STOP>
__device__ __forceinline__ void walk_staged(const StagedRec* __restrict__ st, uint32_t n, float pxf, float pyf, float& T, float& C0,
                                            float& C1, float& C2, uint32_t& last)
{
    StagedRec cur = st[0];
    for (uint32_t k = 0; k < n; k += 2) {
        const StagedRec nxt = st[min(k + 1u, n - 1u)];
        deep_accumulate<STOP>(cur, cur.pos1, pxf, pyf, T, C0, C1, C2, last);
        if (k + 1u >= n) break;
        cur = st[min(k + 2u, n - 1u)];
        deep_accumulate<STOP>(nxt, nxt.pos1, pxf, pyf, T, C0, C1, C2, last);
        if (STOP && __ballot(T > 0.0f) == 0ull) break;
    }
}

// The tile's checkpoint slots as ckpt_begin returns them, without marking them (the tile's own workgroup does that).
__device__ __forceinline__ float4* ckpt_tile_slots(const Ckpt& ck, uint32_t tile, uint32_t n_tile)
{
    if (!ck.state) return nullptr;
    const bool leave = ckpt_leave(ck, n_tile);
    return leave ? ck.state + (size_t)((const_u32p)ck.seg_first)[tile] * 256u : nullptr;
}

// One workgroup (DEEP_LANES waves) = quad q of tile (tx, ty); `n` entries in the quad's compacted list `list` (written by an
// EARLIER kernel: read with plain vector loads).
__device__ __forceinline__ void blend_forward_deep_quad(const Camera& cam, uint32_t lastg, int tx, int ty, int q, uint32_t n,
                                                        const uint64_t* __restrict__ list, const Splat* __restrict__ splats,
                                                        const float* __restrict__ bg, float* __restrict__ out_color,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, int clamp_output,
                                                        float4* __restrict__ ck_tile, uint32_t* __restrict__ nproc_out, DeepShared& sh)
{
    const int lane = threadIdx.x & 63;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int px = tx * TILE + (q & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    float4* ck_mine = ck_tile ? ck_tile + q * 64 + lane : nullptr;
    StagedRec* my_stage = sh.stage[m];

    // per-pixel running state, kept identically by all DEEP_LANES waves (same operations on the same LDS data)
    float T_run = inside ? 1.0f : -1.0f, Cf0 = 0.0f, Cf1 = 0.0f, Cf2 = 0.0f;
    uint32_t last_run = 0, walked = 0;

    // lane l < CKPT_SEG fetches entry (round base + m CKPT_SEG + l) and its record; the entry of the round after next and the
    // record of the next round are in flight while a round is computed
    auto entry_at = [&](uint32_t base) -> uint64_t {
        const uint32_t i = base + m * (uint32_t)CKPT_SEG + (uint32_t)lane;
        return (lane < CKPT_SEG && i < n) ? list[i] : 0ull;
    };
    struct RecRegs { float4 h0, h1; float b; };
    auto record_of = [&](uint64_t e, uint32_t base) -> RecRegs {
        RecRegs r{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), 0.f};
        if (lane < CKPT_SEG && base + m * (uint32_t)CKPT_SEG + (uint32_t)lane < n) {
            const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(splats) + (size_t)min((uint32_t)e & GID_MASK, lastg) * 64u);
            r.h0 = p[0], r.h1 = p[1], r.b = reinterpret_cast<const float*>(p)[8];
        }
        return r;
    };
    HGS_TRACE_PUT(1, wall_clock64());
    HGS_TRACE_PUT(4, n);
    uint64_t ent = entry_at(0);
    RecRegs rec = record_of(ent, 0);
    uint64_t ent_next = entry_at(DEEP_ROUND);

    for (uint32_t base = 0; base < n; base += DEEP_ROUND) {  // (uniform over the workgroup)
        HGS_TRACE_ADD(5, 1);
        const uint32_t buf = (base / DEEP_ROUND) & 1u;
        const uint32_t nseg = min((uint32_t)DEEP_LANES, (n - base + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT);
        const uint32_t off = base + m * (uint32_t)CKPT_SEG;
        const uint32_t cnt = m < nseg ? min((uint32_t)CKPT_SEG, n - off) : 0u;
        // ---- stage this round's records (wave-private: no barrier, the wave's own LDS writes are ordered before its reads)
        if (lane < CKPT_SEG) {
            float4* d = reinterpret_cast<float4*>(&my_stage[lane]);
            d[0] = rec.h0;                                                          // x, y, A, B
            d[1] = rec.h1;                                                          // C, L, r, g
            d[2] = make_float4(rec.b, __uint_as_float((uint32_t)(ent >> 32)), 0.f, 0.f);   // b, pos1
        }
        // ... and fetch ahead: the next round's records (their entries have arrived), the entries of the round after
        ent = ent_next;
        rec = record_of(ent, base + DEEP_ROUND);
        ent_next = entry_at(base + 2u * DEEP_ROUND);
        // ---- phase A: compose my segment from T = 1 -- but wave 0 knows the transmittance its pixels arrive with (T_run): it
        // walks its segment with the exact rule at once and publishes (P, C) = what the segment did to (T, colour) RELATIVE to
        // the start state, so that the scan below treats all segments alike; its pixels that stop inside are final here.
        if (cnt) {
            float P = (m == 0u) ? T_run : 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
            uint32_t ll = 0;
            if (m == 0u) {
                walk_staged<true>(my_stage, cnt, pxf, pyf, P, c0, c1, c2, ll);
            } else
                walk_staged<false>(my_stage, cnt, pxf, pyf, P, c0, c1, c2, ll);
            sh.comp[buf][m][lane] = make_float4(P, c0, c1, c2);   // (wave 0: absolute T after the segment, sign = stopped inside)
            sh.comp_last[buf][m][lane] = ll;
        }
        __syncthreads();
        // ---- phase B: scan over the round's segments
        float t = T_run;
        bool active = t > 0.0f;
        float myT = -1.0f, my0 = 0.0f, my1 = 0.0f, my2 = 0.0f;  // state in front of my segment
        uint32_t myLast = 0;
        bool my_stop = false;
        float4 ckv = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);      // state behind my segment
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)DEEP_LANES; ++j) {
            if (j < nseg) {
                const float4 pc = sh.comp[buf][j][lane];
                const uint32_t lj = sh.comp_last[buf][j][lane];
                if (j == m) myT = active ? t : -1.0f, my0 = Cf0, my1 = Cf1, my2 = Cf2, myLast = last_run;
                // segment 0 was walked exactly from t: pc.x is the transmittance behind it (negative: the pixel stopped inside),
                // its colour is absolute; the others were composed from T = 1: relative
                const float tn = j == 0u ? __builtin_fabsf(pc.x) : t * pc.x;
                const bool stop = active && (j == 0u ? pc.x < 0.0f : tn < T_STOP);
                const bool pass = active && !stop;
                const float wgt = j == 0u ? (active ? 1.0f : 0.0f) : (pass ? t : 0.0f);
                Cf0 = __builtin_fmaf(pc.y, wgt, Cf0);
                Cf1 = __builtin_fmaf(pc.z, wgt, Cf1);
                Cf2 = __builtin_fmaf(pc.w, wgt, Cf2);
                last_run = ((j == 0u ? active : pass) && lj != 0u) ? lj : last_run;
                t = (j == 0u ? active : pass) ? tn : t;
                active = pass;
                if (j == m) my_stop = stop, ckv = make_float4(pass ? tn : -__builtin_fabsf(t), Cf0, Cf1, Cf2);
            }
        }
        T_run = active ? t : -__builtin_fabsf(t);
        walked = min(n, base + DEEP_ROUND);
        const bool quad_done = __ballot(T_run > 0.0f) == 0ull || walked == n;  // (the same in all waves of the workgroup)
        // ---- the pixels that stop inside my segment: exact walk from the transmittance they arrive with
        if (m == 0u) {
            // (wave 0 walked exactly: the pixels that stopped inside its segment are final with the state the walk left --
            //  the scan above has already added the segment's colour for them and set last_run / t to the walk's results)
            if (my_stop) {
                sh.fin[lane] = ckv;
                sh.fin_last[lane] = last_run;
            }
        } else if (cnt && __ballot(my_stop) != 0ull) {
            HGS_TRACE_ADD(6, 1);
            float T = my_stop ? myT : -1.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
            uint32_t rl = 0;
            walk_staged<true>(my_stage, cnt, pxf, pyf, T, r0, r1, r2, rl);
            if (my_stop) {
                // (a pixel the exact products leave a hair above 1e-4 is done all the same: its next contributing entry would
                //  stop it without being added)
                ckv = make_float4(-__builtin_fabsf(T), my0 + r0, my1 + r1, my2 + r2);
                sh.fin[lane] = ckv;
                sh.fin_last[lane] = rl ? rl : myLast;
            }
        }
        if (ck_mine && cnt) {
            // slot k = state BEHIND segment k; the last slot the quad uses receives the end state below instead
            const uint32_t k = (base >> CKPT_SHIFT) + m, k_last = ((walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT) - 1u;
            if (!(quad_done && k == k_last)) ck_mine[(size_t)k * 256u] = ckv;
        }
        if (quad_done) break;
    }
    __syncthreads();  // the stopped pixels' final states are in LDS
    if (m == 0u) {
        float T = T_run, c0 = Cf0, c1 = Cf1, c2 = Cf2;
        uint32_t last = last_run;
        if (inside && T_run < 0.0f) {
            const float4 f = sh.fin[lane];
            T = f.x, c0 = f.y, c1 = f.z, c2 = f.w, last = sh.fin_last[lane];
        }
        if (ck_tile) {
            const uint32_t segs = (walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT;
            if (segs >= 2u) ck_mine[(size_t)(segs - 1u) * 256u] = make_float4(T, c0, c1, c2);
            if (lane == 0) *nproc_out = walked;
        }
        if (inside) write_pixel(cam, px, py, T, c0, c1, c2, last, bg, out_color, final_T, n_contrib, clamp_output);
    }
    __syncthreads();  // LDS is free for the workgroup's next quad
    HGS_TRACE_PUT(2, wall_clock64());
    HGS_TRACE_ADD(7, 1);
}

// Workgroup `worker` of `num_workers` (DEEP_LANES waves each) takes every num_workers-th (long tile, quad) pair of the
// device-built long-tile list; candidates that are not long on this frame (n <= n_total[4]) belong to the per-tile workgroups.
// mark_slots: the workers also record which tile their checkpoint slots belong to (ckpt_begin) -- when no per-tile workgroup
// of the same launch does it.
__device__ DEEP_INLINE void deep_forward_worker(uint32_t worker, uint32_t num_workers, const Camera& cam, uint32_t lastg,
                                                    const uint2* __restrict__ ranges, const uint64_t* __restrict__ act, size_t stride,
                                                    const uint32_t* __restrict__ act_count, const Splat* __restrict__ splats,
                                                    const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ final_T,
                                                    uint32_t* __restrict__ n_contrib, int clamp_output, const Ckpt& ck,
                                                    const uint32_t* __restrict__ large_tiles, const uint32_t* __restrict__ n_total, DeepShared& sh,
                                                    bool mark_slots)
{
    // (long tiles of at most n_total[8] entries are blended one wave per quad by whoever blends the other tiles)
    const uint32_t count = 4u * ((const_u32p)n_total)[2], threshold = max(((const_u32p)n_total)[4], ((const_u32p)n_total)[8]);
    for (uint32_t item = worker; item < count; item += num_workers) {
        const uint32_t tile = ((const_u32p)large_tiles)[item >> 2], q = item & 3u;
        const v2u rg = ((const_u2p)ranges)[tile];
        const uint32_t n_tile = rg.y - rg.x;
        if (n_tile <= threshold) continue;
        // (no per-tile workgroup runs beside the workers in the repair pass: quad 0's workgroup marks the tile's checkpoint slots)
        if (mark_slots && q == 0u) ckpt_begin(ck, tile, n_tile);
        const uint32_t nq = ((const_u32p)act_count)[tile * NUM_LISTS + q];
        blend_forward_deep_quad(cam, lastg, (int)(tile % (uint32_t)cam.gx), (int)(tile / (uint32_t)cam.gx), (int)q, nq,
                                act + (size_t)q * stride + rg.x, splats, bg, out_color, final_T, n_contrib, clamp_output,
                                ckpt_tile_slots(ck, tile, n_tile), ck.quad_nproc + tile * 4u + q, sh);
    }
}

}  // namespace hgs
