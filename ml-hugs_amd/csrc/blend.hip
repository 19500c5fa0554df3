// Per-tile alpha blending: forward (K6) and pixel-side backward (K7).  SURVEY.md A.4 / A.5.
//
// CDNA4 design (not the CUDA shape):
//  * A 16x16 tile is one 256-thread workgroup = 4 independent wave64s, each owning an 8x8 pixel quad.
//    There is NO LDS staging and NO barrier: the splat record of a list entry is the same for every lane,
//    so it is fetched through the scalar data cache (s_load_dwordx8 from the constant address space)
//    straight into SGPRs and used as a scalar operand of the VALU math.
//  * Each quad has a bitmap over the sorted list (built by the tile-ranges kernel from the coverage masks
//    the emission kernel computed).  A wave walks only the set bits of its bitmap with scalar bit scans
//    (s_ff1 / s_flbit), so list entries that cannot touch its 64 pixels cost no vector instruction at all
//    -- in the 200k / 1080p workload that is ~60 % of all (wave, splat) pairs.
//  * The walk is software pipelined: the list value of entry n+2 and the record of entry n+1 are in
//    flight while entry n is blended out of SGPRs (scalar loads return out of order, so the single
//    lgkmcnt(0) sits at the top of the iteration, before the next record load is issued).
//  * The per-pixel update is fully predicated (v_cndmask), early-out is per wave (64 pixels).
//  * Backward: the 9 per-splat partial sums of a wave are combined with a butterfly transpose-reduce
//    (quad_perm / row_shl / row_shr / row_ror DPP + two cross-row shuffles) that leaves the totals in 9
//    different lanes, which then issue ONE global_atomic_add_f32 instruction into a [P][12] accumulator
//    record (contiguous 36 bytes per Gaussian).
//
// Compiled with -ffp-contract=off; the FMAs below are explicit so forward and backward evaluate alpha
// with the identical instruction sequence (backward must re-take forward's skip decisions).
#include "hgs_common.h"

namespace hgs {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v4f* const_f4p;
typedef const __attribute__((address_space(4))) float* const_f32p;
typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
typedef const __attribute__((address_space(4))) uint64_t* const_u64p;
typedef const __attribute__((address_space(4))) v2u* const_u2p;

struct SplatRec {  // wave-uniform (lives in SGPRs)
    float x, y, A, B, C, op, r, g, b;
};

__device__ __forceinline__ SplatRec load_rec(const Splat* splats, uint32_t gid)
{
    // 32-bit byte offset (P * 48 < 2^32 is checked by the API): one s_mul_i32 + base+offset scalar loads
    const_f4p p = (const_f4p)((const char*)splats + gid * 48u);
    const v4f h0 = p[0], h1 = p[1];
    const float b = ((const_f32p)p)[8];
    SplatRec s;
    s.x = h0.x, s.y = h0.y;
    s.A = h0.z, s.B = h0.w, s.C = h1.x;  // half-conic form: power = A dx^2 + B dx dy + C dy^2
    s.op = h1.y, s.r = h1.z, s.g = h1.w, s.b = b;
    return s;
}

__device__ __forceinline__ float gauss_power(const SplatRec& s, float dx, float dy)
{
    float t = __builtin_fmaf(s.A, dx, s.B * dy);
    float u = s.C * dy;
    return __builtin_fmaf(dx, t, u * dy);
}

// XCD-aware tile order: consecutive workgroup ids are dealt round-robin to the 8 XCDs, so give each
// XCD a contiguous band of tiles (neighbouring tiles share splats => they hit the same L2).
__device__ __forceinline__ int remap_tile(int bid, int num_tiles)
{
    const int q = num_tiles / 8, r = num_tiles % 8;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Scalar walk over the set bits of one quad bitmap restricted to list positions [s, e).
struct BitWalk {
    const_u64p bm;
    uint32_t s, e;  // slice of the sorted list
    uint32_t wd;    // current word
    uint64_t m;     // unvisited bits of the current word

    __device__ __forceinline__ uint64_t word(uint32_t w) const
    {
        uint64_t v = bm[w];
        if (w == (s >> 6)) v &= ~0ull << (s & 63u);
        if (w == ((e - 1u) >> 6)) {
            const uint32_t r = e & 63u;
            if (r) v &= (1ull << r) - 1ull;
        }
        return v;
    }
    __device__ __forceinline__ void begin_up(const_u64p bm_, uint32_t s_, uint32_t e_)
    {
        bm = bm_, s = s_, e = e_, wd = s_ >> 6;
        m = word(wd);
    }
    __device__ __forceinline__ bool next_up(uint32_t& i)  // ascending
    {
        const uint32_t last = (e - 1u) >> 6;
        while (m == 0ull) {
            if (wd >= last) return false;
            ++wd;
            m = word(wd);
        }
        const uint32_t k = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        i = (wd << 6) + k;
        return true;
    }
    __device__ __forceinline__ void begin_down(const_u64p bm_, uint32_t s_, uint32_t e_)
    {
        bm = bm_, s = s_, e = e_, wd = (e_ - 1u) >> 6;
        m = word(wd);
    }
    __device__ __forceinline__ bool next_down(uint32_t& i)  // descending
    {
        const uint32_t first = s >> 6;
        while (m == 0ull) {
            if (wd <= first) return false;
            --wd;
            m = word(wd);
        }
        const uint32_t k = 63u - (uint32_t)__builtin_clzll(m);
        m &= ~(1ull << k);
        i = (wd << 6) + k;
        return true;
    }
};

// ------------------------------------------------------------------------------------------------
// One list entry applied to the wave's 64 pixels, fully predicated (v_cndmask, no exec-mask branches);
// `pos1` is the entry's 1-based position in the tile list (wave-uniform).
__device__ __forceinline__ void fwd_accumulate(const SplatRec& s, uint32_t pos1, float pxf, float pyf, float& T,
                                               float& C0, float& C1, float& C2, uint32_t& last, bool& done)
{
    const float dx = s.x - pxf, dy = s.y - pyf;
    const float power = gauss_power(s, dx, dy);
    const float alpha = fminf(ALPHA_MAX, s.op * __expf(power));
    const bool ok = !done && power <= 0.0f && alpha >= ALPHA_MIN;
    const float test_T = T * (1.0f - alpha);
    const bool stop = ok && test_T < T_STOP;
    const bool upd = ok && !stop;
    const float wgt = upd ? alpha * T : 0.0f;
    C0 = __builtin_fmaf(s.r, wgt, C0);
    C1 = __builtin_fmaf(s.g, wgt, C1);
    C2 = __builtin_fmaf(s.b, wgt, C2);
    T = upd ? test_T : T;
    last = upd ? pos1 : last;
    done = done || stop;
}

__global__ void __launch_bounds__(256)
blend_forward_kernel(Camera cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const uint64_t* __restrict__ bitmaps, uint32_t bitmap_words, const Splat* __restrict__ splats,
                     const float* __restrict__ bg, float* __restrict__ out_color, float* __restrict__ final_T,
                     uint32_t* __restrict__ n_contrib)
{
    const int tile = remap_tile(blockIdx.x, cam.gx * cam.gy);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const v2u range = ((const_u2p)ranges)[tile];
    const_u32p list = (const_u32p)point_list;

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    bool done = !inside;

    if (range.y > range.x) {
        BitWalk it;
        it.begin_up((const_u64p)bitmaps + (size_t)w * bitmap_words, range.x, range.y);
        uint32_t i_cur = 0, i_nxt = 0, i_3 = 0;
        if (it.next_up(i_cur)) {
            const uint32_t val_cur = list[i_cur];
            bool v_nxt = it.next_up(i_nxt);
            uint32_t val_nxt = list[v_nxt ? i_nxt : i_cur];
            SplatRec rec_cur = load_rec(splats, val_cur & GID_MASK);
            while (true) {
                // top of the pipeline: everything issued one iteration ago has had a full blend to land
                const SplatRec rec_nxt = load_rec(splats, val_nxt & GID_MASK);
                const bool v_3 = v_nxt && it.next_up(i_3);
                const uint32_t val_3 = list[v_3 ? i_3 : i_cur];
                fwd_accumulate(rec_cur, i_cur - range.x + 1u, pxf, pyf, T, C0, C1, C2, last, done);
                if (!v_nxt || __ballot(!done) == 0ull) break;
                rec_cur = rec_nxt;
                i_cur = i_nxt, i_nxt = i_3;
                v_nxt = v_3, val_nxt = val_3;
            }
        }
    }
    if (inside) {
        const size_t HW = (size_t)cam.H * cam.W, pix = (size_t)py * cam.W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = __builtin_fmaf(T, bg[0], C0);
        out_color[HW + pix] = __builtin_fmaf(T, bg[1], C1);
        out_color[2 * HW + pix] = __builtin_fmaf(T, bg[2], C2);
    }
}

void launch_blend_forward(const Camera& cam, const uint2* ranges, const uint32_t* point_list, const uint64_t* bitmaps,
                          size_t bitmap_words, const Splat* splats, const float* bg, float* out_color, float* final_T,
                          uint32_t* n_contrib, hipStream_t st)
{
    hipLaunchKernelGGL(blend_forward_kernel, dim3(cam.gx * cam.gy), dim3(256), 0, st, cam, ranges, point_list, bitmaps,
                       (uint32_t)bitmap_words, splats, bg, out_color, final_T, n_contrib);
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
constexpr int DPP_ROW_SHL4 = 0x104;  // lane i <- lane i+4 (within a row of 16)
constexpr int DPP_ROW_SHR4 = 0x114;  // lane i <- lane i-4
constexpr int DPP_ROW_ROR8 = 0x128;  // lane i <- lane i^8 (rotate by half a row)

// pairwise transpose-reduce step: afterwards lanes with `hi` clear hold (a + partner's a) and lanes
// with `hi` set hold (b + partner's b); partner = lane ^ XOR within the quad.
template <int CTRL>
__device__ __forceinline__ float pair_step(float a, float b, bool hi)
{
    const float keep = hi ? b : a, send = hi ? a : b;
    return keep + dpp_mov<CTRL>(send);
}

__global__ void __launch_bounds__(256)
blend_backward_kernel(Camera cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint64_t* __restrict__ bitmaps, uint32_t bitmap_words, const Splat* __restrict__ splats,
                      const float* __restrict__ bg, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                      float* __restrict__ grad_accum)
{
    const int tile = remap_tile(blockIdx.x, cam.gx * cam.gy);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const v2u range = ((const_u2p)ranges)[tile];
    const_u32p list = (const_u32p)point_list;
    const size_t HW = (size_t)cam.H * cam.W, pix = (size_t)py * cam.W + px;

    const float T_final = inside ? final_T[pix] : 0.0f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dpix[pix] : 0.0f;
    const float g1 = inside ? dL_dpix[HW + pix] : 0.0f;
    const float g2 = inside ? dL_dpix[2 * HW + pix] : 0.0f;
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    const float ddelx_dx = 0.5f * (float)cam.W, ddely_dy = 0.5f * (float)cam.H;

    // the wave starts at the deepest entry any of its pixels composited
    uint32_t wmax = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    if (wmax == 0) return;

    float ar0 = 0.0f, ar1 = 0.0f, ar2 = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f, last_alpha = 0.0f;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;

    BitWalk it;
    it.begin_down((const_u64p)bitmaps + (size_t)w * bitmap_words, range.x, range.x + wmax);
    uint32_t i_cur = 0, i_nxt = 0, i_3 = 0;
    if (!it.next_down(i_cur)) return;
    uint32_t val_cur = list[i_cur];
    bool v_nxt = it.next_down(i_nxt);
    uint32_t val_nxt = list[v_nxt ? i_nxt : i_cur];
    SplatRec s = load_rec(splats, val_cur & GID_MASK);
    while (true) {
        const SplatRec rec_nxt = load_rec(splats, val_nxt & GID_MASK);
        const bool v_3 = v_nxt && it.next_down(i_3);
        const uint32_t val_3 = list[v_3 ? i_3 : i_cur];

        const uint32_t pos1 = i_cur - range.x + 1u;
        const float dx = s.x - pxf, dy = s.y - pyf;
        const float power = gauss_power(s, dx, dy);
        const float G = __expf(power);
        const float alpha = fminf(ALPHA_MAX, s.op * G);
        const bool act = pos1 <= last_contributor && power <= 0.0f && alpha >= ALPHA_MIN;
        if (__ballot(act) != 0ull) {
            float v_mx = 0.0f, v_my = 0.0f, v_cxx = 0.0f, v_cxy = 0.0f, v_cyy = 0.0f, v_op = 0.0f, v_r = 0.0f,
                  v_g = 0.0f, v_b = 0.0f;
            if (act) {
                const float one_m = 1.0f - alpha;
                const float inv = __builtin_amdgcn_rcpf(one_m);
                T = T * inv;
                const float dch = alpha * T;
                ar0 = __builtin_fmaf(last_alpha, lc0, (1.0f - last_alpha) * ar0);
                ar1 = __builtin_fmaf(last_alpha, lc1, (1.0f - last_alpha) * ar1);
                ar2 = __builtin_fmaf(last_alpha, lc2, (1.0f - last_alpha) * ar2);
                lc0 = s.r, lc1 = s.g, lc2 = s.b;
                float dL_dalpha = (s.r - ar0) * g0;
                dL_dalpha = __builtin_fmaf(s.g - ar1, g1, dL_dalpha);
                dL_dalpha = __builtin_fmaf(s.b - ar2, g2, dL_dalpha);
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha = __builtin_fmaf(-T_final * inv, bg_dot, dL_dalpha);
                const float dL_dG = s.op * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                // conic = (-2A, -B, -2C)
                const float dG_ddelx = 2.0f * gdx * s.A + gdy * s.B;
                const float dG_ddely = 2.0f * gdy * s.C + gdx * s.B;
                v_mx = dL_dG * dG_ddelx * ddelx_dx;
                v_my = dL_dG * dG_ddely * ddely_dy;
                const float h = -0.5f * dL_dG;
                v_cxx = h * gdx * dx;
                v_cxy = h * gdx * dy;
                v_cyy = h * gdy * dy;
                v_op = G * dL_dalpha;
                v_r = dch * g0, v_g = dch * g1, v_b = dch * g2;
            }
            // ---- butterfly transpose-reduce of 8 values; lane (l & 7) == k ends up owning value k ----
            // slot order k: 0 mx, 1 my, 2 cxx, 3 cxy, 4 cyy, 5 op, 6 r, 7 g   (+ b reduced on its own -> lane 8)
            const float w0 = pair_step<DPP_QUAD_XOR1>(v_mx, v_my, b0);
            const float w1 = pair_step<DPP_QUAD_XOR1>(v_cxx, v_cxy, b0);
            const float w2 = pair_step<DPP_QUAD_XOR1>(v_cyy, v_op, b0);
            const float w3 = pair_step<DPP_QUAD_XOR1>(v_r, v_g, b0);
            const float x0 = pair_step<DPP_QUAD_XOR2>(w0, w1, b1);
            const float x1 = pair_step<DPP_QUAD_XOR2>(w2, w3, b1);
            const float x1_dn = dpp_mov<DPP_ROW_SHR4>(x1), x0_up = dpp_mov<DPP_ROW_SHL4>(x0);
            float y = b2 ? (x1 + x1_dn) : (x0 + x0_up);
            float vb = v_b + dpp_mov<DPP_QUAD_XOR1>(v_b);
            vb += dpp_mov<DPP_QUAD_XOR2>(vb);
            const float vb_dn = dpp_mov<DPP_ROW_SHR4>(vb), vb_up = dpp_mov<DPP_ROW_SHL4>(vb);
            vb += b2 ? vb_dn : vb_up;
            y += dpp_mov<DPP_ROW_ROR8>(y);
            vb += dpp_mov<DPP_ROW_ROR8>(vb);
            y += __shfl_xor(y, 16, 64);
            vb += __shfl_xor(vb, 16, 64);
            y += __shfl_xor(y, 32, 64);
            vb += __shfl_xor(vb, 32, 64);
            // lanes 0..8 add the nine totals into the Gaussian's accumulator record with one atomic instruction
            if (lane < 9) atomicAdd(grad_accum + (size_t)(val_cur & GID_MASK) * 12u + lane, lane == 8 ? vb : y);
        }
        if (!v_nxt) break;
        s = rec_nxt;
        val_cur = val_nxt;
        i_cur = i_nxt, i_nxt = i_3;
        v_nxt = v_3, val_nxt = val_3;
    }
}

void launch_blend_backward(const Camera& cam, const uint2* ranges, const uint32_t* point_list, const uint64_t* bitmaps,
                           size_t bitmap_words, const Splat* splats, const float* bg, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, float* grad_accum, hipStream_t st)
{
    hipLaunchKernelGGL(blend_backward_kernel, dim3(cam.gx * cam.gy), dim3(256), 0, st, cam, ranges, point_list, bitmaps,
                       (uint32_t)bitmap_words, splats, bg, final_T, n_contrib, dL_dpix, grad_accum);
}

}  // namespace hgs
