// Per-tile alpha blending: forward (K6) and pixel-side backward (K7).  SURVEY.md A.4 / A.5.
//
// CDNA4 design (not the CUDA shape):
//  * A 16x16 tile is one 256-thread workgroup = 4 independent wave64s, each owning an 8x8 pixel
//    quad.  There is NO LDS staging and NO barrier: the splat record of list entry j is the same
//    for every lane, so it is fetched through the scalar data cache (s_load_dwordx4 from the
//    constant address space) straight into SGPRs and used as a scalar operand of the VALU math.
//  * Early-out is per wave (64 pixels), not per 256-pixel tile.
//  * Backward: the 9 per-splat partial sums of a wave are combined with a butterfly
//    transpose-reduce (quad_perm / row_shl / row_shr / row_ror DPP + two cross-row shuffles) that
//    leaves 8 of the totals in 8 different lanes, so one global_atomic_add_f32 instruction retires
//    8 of the 9 accumulations; wave-splats that no pixel touches are skipped with one ballot.
//
// Compiled with -ffp-contract=off; the FMAs below are explicit so forward and backward evaluate
// alpha with the identical instruction sequence (backward must re-take forward's skip decisions).
#include "hgs_common.h"

namespace hgs {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v4f* const_f4p;
typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
typedef const __attribute__((address_space(4))) v2u* const_u2p;

struct SplatRec {  // wave-uniform (lives in SGPRs)
    float x, y, A, B, C, op, r, g, b;
};

__device__ __forceinline__ SplatRec load_rec(const Splat* splats, uint32_t gid)
{
    const_f4p p = (const_f4p)(splats + gid);
    const v4f h0 = p[0], h1 = p[1];
    float b = ((const __attribute__((address_space(4))) float*)p)[8];
    SplatRec s;
    s.x = h0.x, s.y = h0.y;
    // half-conic form: power = A dx^2 + B dx dy + C dy^2  (exact power-of-two rescale of the conic)
    s.A = -0.5f * h0.z, s.B = -h0.w, s.C = -0.5f * h1.x;
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

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
blend_forward_kernel(Camera cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                     const Splat* __restrict__ splats, const float* __restrict__ bg, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    const int tile = remap_tile(blockIdx.x, cam.gx * cam.gy);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const v2u range = ((const_u2p)ranges)[tile];
    const_u32p list = (const_u32p)point_list;

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;

    for (uint32_t j0 = range.x; j0 < range.y; j0 += 4) {
        if (__ballot(!done) == 0ull) break;
        // four list entries per trip: index loads, then record loads, are issued back to back
        uint32_t gid[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gid[k] = list[j0 + k];  // list is padded: reading past range.y is safe
        SplatRec rec[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[k] = load_rec(splats, j0 + k < range.y ? gid[k] : gid[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (j0 + k < range.y && !done) {
                const SplatRec& s = rec[k];
                ++contributor;
                const float dx = s.x - pxf, dy = s.y - pyf;
                const float power = gauss_power(s, dx, dy);
                if (power <= 0.0f) {
                    const float alpha = fminf(ALPHA_MAX, s.op * __expf(power));
                    if (alpha >= ALPHA_MIN) {
                        const float test_T = T * (1.0f - alpha);
                        if (test_T < T_STOP) {
                            done = true;
                        } else {
                            const float wgt = alpha * T;
                            C0 = __builtin_fmaf(s.r, wgt, C0);
                            C1 = __builtin_fmaf(s.g, wgt, C1);
                            C2 = __builtin_fmaf(s.b, wgt, C2);
                            T = test_T;
                            last = contributor;
                        }
                    }
                }
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

void launch_blend_forward(const Camera& cam, const uint2* ranges, const uint32_t* point_list, const Splat* splats,
                          const float* bg, float* out_color, float* final_T, uint32_t* n_contrib, hipStream_t st)
{
    hipLaunchKernelGGL(blend_forward_kernel, dim3(cam.gx * cam.gy), dim3(256), 0, st, cam, ranges, point_list, splats,
                       bg, out_color, final_T, n_contrib);
}

// ------------------------------------------------------------------------------------------------
// DPP helpers
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

// pairwise transpose-reduce step: after it, lanes with `hi` clear hold (a + partner's a) and lanes
// with `hi` set hold (b + partner's b); partner = lane ^ XOR within the quad.
template <int CTRL>
__device__ __forceinline__ float pair_step(float a, float b, bool hi)
{
    const float keep = hi ? b : a, send = hi ? a : b;
    return keep + dpp_mov<CTRL>(send);
}

__global__ void __launch_bounds__(256)
blend_backward_kernel(Camera cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const Splat* __restrict__ splats, const float* __restrict__ bg,
                      const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D,
                      float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors)
{
    const int tile = remap_tile(blockIdx.x, cam.gx * cam.gy);
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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

    float ar0 = 0.0f, ar1 = 0.0f, ar2 = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f, last_alpha = 0.0f;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;

    for (uint32_t c = wmax; c >= 1; --c) {  // c = 1-based position in the tile list
        const uint32_t gid = list[range.x + c - 1];
        const SplatRec s = load_rec(splats, gid);
        const float dx = s.x - pxf, dy = s.y - pyf;
        const float power = gauss_power(s, dx, dy);
        const float G = __expf(power);
        const float alpha = fminf(ALPHA_MAX, s.op * G);
        const bool act = c <= last_contributor && power <= 0.0f && alpha >= ALPHA_MIN;
        if (__ballot(act) == 0ull) continue;

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
#ifdef HGS_BWD_SIMPLE_REDUCE
        {
            float vals[9] = {v_mx, v_my, v_cxx, v_cxy, v_cyy, v_op, v_r, v_g, v_b};
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) vals[q] += __shfl_xor(vals[q], d, 64);
            if (lane == 0) {
                atomicAdd(dL_dmean2D + 3 * (size_t)gid, vals[0]);
                atomicAdd(dL_dmean2D + 3 * (size_t)gid + 1, vals[1]);
                atomicAdd(dL_dconic + 4 * (size_t)gid, vals[2]);
                atomicAdd(dL_dconic + 4 * (size_t)gid + 1, vals[3]);
                atomicAdd(dL_dconic + 4 * (size_t)gid + 3, vals[4]);
                atomicAdd(dL_dopacity + gid, vals[5]);
                atomicAdd(dL_dcolors + 3 * (size_t)gid, vals[6]);
                atomicAdd(dL_dcolors + 3 * (size_t)gid + 1, vals[7]);
                atomicAdd(dL_dcolors + 3 * (size_t)gid + 2, vals[8]);
            }
            continue;
        }
#endif
        // ---- butterfly transpose-reduce of 8 values; lane (l & 7) == k ends up owning value k ----
        // value order k: 0 mx, 1 my, 2 cxx, 3 cxy, 4 cyy, 5 op, 6 r, 7 g   (+ b reduced on its own)
        float w0 = pair_step<DPP_QUAD_XOR1>(v_mx, v_my, b0);
        float w1 = pair_step<DPP_QUAD_XOR1>(v_cxx, v_cxy, b0);
        float w2 = pair_step<DPP_QUAD_XOR1>(v_cyy, v_op, b0);
        float w3 = pair_step<DPP_QUAD_XOR1>(v_r, v_g, b0);
        float x0 = pair_step<DPP_QUAD_XOR2>(w0, w1, b1);
        float x1 = pair_step<DPP_QUAD_XOR2>(w2, w3, b1);
        // NB: every DPP move is evaluated unconditionally (all 64 lanes active) and only then selected;
        // inside a ?: arm the compiler would run it under a partial exec mask and read dead lanes.
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
        // lanes 0..7 scatter the 8 totals, lane 8 the ninth
        if (lane < 9) {
            float* dst;
            switch (lane) {
                case 0: dst = dL_dmean2D + 3 * (size_t)gid; break;
                case 1: dst = dL_dmean2D + 3 * (size_t)gid + 1; break;
                case 2: dst = dL_dconic + 4 * (size_t)gid; break;
                case 3: dst = dL_dconic + 4 * (size_t)gid + 1; break;
                case 4: dst = dL_dconic + 4 * (size_t)gid + 3; break;
                case 5: dst = dL_dopacity + gid; break;
                case 6: dst = dL_dcolors + 3 * (size_t)gid; break;
                case 7: dst = dL_dcolors + 3 * (size_t)gid + 1; break;
                default: dst = dL_dcolors + 3 * (size_t)gid + 2; break;
            }
            atomicAdd(dst, lane == 8 ? vb : y);
        }
    }
}

void launch_blend_backward(const Camera& cam, const uint2* ranges, const uint32_t* point_list, const Splat* splats,
                           const float* bg, const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                           float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, hipStream_t st)
{
    hipLaunchKernelGGL(blend_backward_kernel, dim3(cam.gx * cam.gy), dim3(256), 0, st, cam, ranges, point_list, splats,
                       bg, final_T, n_contrib, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors);
}

}  // namespace hgs
