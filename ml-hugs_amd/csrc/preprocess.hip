// Per-Gaussian kernels: forward projection (K1), fused per-Gaussian backward (K8+K9), markVisible (K10).
// Algorithm: SURVEY.md Appendix A.2 / A.3 / A.5 (the arithmetic the reference obtains from
// diff_gaussian_rasterization at /root/reference/hugs/renderer/gs_renderer.py:144-152).
//
// This translation unit is compiled with -ffp-contract=off: radius, tile rectangle, tiles_touched and
// the depth bits are integer functions of fp32 arithmetic and must round exactly as written (they are
// compared bit-for-bit against the CPU oracle).  Division and sqrt are IEEE (correctly rounded).
#include <cstdlib>

#include "hgs_common.h"
#include "binning_walk.h"

namespace hgs {

__device__ __forceinline__ void sh_basis(int D, float x, float y, float z, float* B)
{
    B[0] = (float)0.28209479177387814;
    if (D > 0) {
        const float c1 = (float)0.4886025119029199;
        B[1] = -c1 * y;
        B[2] = c1 * z;
        B[3] = -c1 * x;
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = (float)1.0925484305920792 * xy;
            B[5] = (float)-1.0925484305920792 * yz;
            B[6] = (float)0.31539156525252005 * (2.0f * zz - xx - yy);
            B[7] = (float)-1.0925484305920792 * xz;
            B[8] = (float)0.5462742152960396 * (xx - yy);
            if (D > 2) {
                B[9] = (float)-0.5900435899266435 * y * (3.0f * xx - yy);
                B[10] = (float)2.890611442640554 * xy * z;
                B[11] = (float)-0.4570457994644658 * y * (4.0f * zz - xx - yy);
                B[12] = (float)0.3731763325901154 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = (float)-0.4570457994644658 * x * (4.0f * zz - xx - yy);
                B[14] = (float)1.445305721320277 * z * (xx - yy);
                B[15] = (float)-0.5900435899266435 * x * (xx - 3.0f * yy);
            }
        }
    }
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* __restrict__ sc, float mod,
                                                     const float* __restrict__ q, float* S)
{
    float s0 = mod * sc[0], s1 = mod * sc[1], s2 = mod * sc[2];
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - r * z), R02 = 2.0f * (x * z + r * y);
    float R10 = 2.0f * (x * y + r * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - r * x);
    float R20 = 2.0f * (x * z - r * y), R21 = 2.0f * (y * z + r * x), R22 = 1.0f - 2.0f * (x * x + y * y);
    float M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
    float M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
    float M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
    S[0] = M00 * M00 + M01 * M01 + M02 * M02;
    S[1] = M00 * M10 + M01 * M11 + M02 * M12;
    S[2] = M00 * M20 + M01 * M21 + M02 * M22;
    S[3] = M10 * M10 + M11 * M11 + M12 * M12;
    S[4] = M10 * M20 + M11 * M21 + M12 * M22;
    S[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

struct Ewa {
    float tx, ty, tz;
    bool x_in, y_in;
    float T00, T01, T02, T10, T11, T12;
    float a, b, c;
};

// EWA projection of Sigma3D to screen space (A.2 step 4); shared by forward and backward.
__device__ __forceinline__ void ewa_project(const float* pv, const Camera& cam, const float* __restrict__ V,
                                            const float* S, Ewa& e)
{
    float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
    float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e.x_in = !(txtz < -limx || txtz > limx);
    e.y_in = !(tytz < -limy || tytz > limy);
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
    e.tz = pv[2];
    float J00 = cam.fx / e.tz;
    float J02 = -(cam.fx * e.tx) / (e.tz * e.tz);
    float J11 = cam.fy / e.tz;
    float J12 = -(cam.fy * e.ty) / (e.tz * e.tz);
    e.T00 = J00 * V[0] + J02 * V[2];
    e.T01 = J00 * V[4] + J02 * V[6];
    e.T02 = J00 * V[8] + J02 * V[10];
    e.T10 = J11 * V[1] + J12 * V[2];
    e.T11 = J11 * V[5] + J12 * V[6];
    e.T12 = J11 * V[9] + J12 * V[10];
    float u00 = S[0] * e.T00 + S[1] * e.T01 + S[2] * e.T02;
    float u01 = S[1] * e.T00 + S[3] * e.T01 + S[4] * e.T02;
    float u02 = S[2] * e.T00 + S[4] * e.T01 + S[5] * e.T02;
    float u10 = S[0] * e.T10 + S[1] * e.T11 + S[2] * e.T12;
    float u11 = S[1] * e.T10 + S[3] * e.T11 + S[4] * e.T12;
    float u12 = S[2] * e.T10 + S[4] * e.T11 + S[5] * e.T12;
    e.a = (e.T00 * u00 + e.T01 * u01 + e.T02 * u02) + 0.3f;
    e.b = e.T00 * u10 + e.T01 * u11 + e.T02 * u12;
    e.c = (e.T10 * u10 + e.T11 * u11 + e.T12 * u12) + 0.3f;
}

template <int K>
__device__ __forceinline__ void sh_backward_rows(const float (&B)[16], const float* sh, float* dsh,  // (sh may BE dsh: an LDS row)
                                                 float dr0, float dr1, float dr2, float (&shw)[16])
{
    // (round 6, measured and closed) all 48 coefficients of degree 3 in flight at once, next to the 16 basis values and the 16 dot products,
    // are what holds the kernel at 128 VGPRs = four waves per SIMD.  Fetched eight (four) at a time -- -DHGS_K8_SH_CHUNK=8 -DHGS_K8_WAVES=5
    // -- it fits 96 VGPRs with 8 (4) spilled and runs SLOWER: C2 29.5 -> 33-35 us, 1 M Gaussians 114 -> 128-132 us (six waves: 44 / 166 us).
    // The kernel is a chain of dependent round trips; what hides them is the loads a THREAD has in flight, not a fifth wave
    // (DESIGN_HISTORY.md, round 6).  16 = one pass.
#ifndef HGS_K8_SH_CHUNK
#define HGS_K8_SH_CHUNK 16
#endif
    constexpr int CH = HGS_K8_SH_CHUNK;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += CH) {
        constexpr int dummy = 0;
        (void)dummy;
        float c[CH][3];
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k0 + k < K) c[k][0] = sh[3 * (k0 + k)], c[k][1] = sh[3 * (k0 + k) + 1], c[k][2] = sh[3 * (k0 + k) + 2];
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k0 + k < K) {
                shw[k0 + k] = c[k][0] * dr0 + c[k][1] * dr1 + c[k][2] * dr2;
                dsh[3 * (k0 + k)] = B[k0 + k] * dr0, dsh[3 * (k0 + k) + 1] = B[k0 + k] * dr1, dsh[3 * (k0 + k) + 2] = B[k0 + k] * dr2;
            }
        if (k0 + CH < K) __builtin_amdgcn_sched_barrier(0);   // (the next chunk's loads stay behind this one's use)
    }
#pragma unroll
    for (int k = K; k < 16; ++k) shw[k] = 0.0f;
}

// colour = sum_k B[k] * sh[k], the same left-to-right summation order for every K (bit-exact against the oracle)
template <int K>
__device__ __forceinline__ void sh_dot(const float (&B)[16], const float* __restrict__ sh, float& a0, float& a1, float& a2)
{
    float c[K][3];
#pragma unroll
    for (int k = 0; k < K; ++k) c[k][0] = sh[3 * k], c[k][1] = sh[3 * k + 1], c[k][2] = sh[3 * k + 2];
#pragma unroll
    for (int k = 0; k < K; ++k) a0 += B[k] * c[k][0], a1 += B[k] * c[k][1], a2 += B[k] * c[k][2];
}

// Round 5: the SH rows through LDS.  A thread reading its own 192-byte row makes every load instruction of the wave touch 64
// different lines (one dword or one float4 of each); the wave's 64 rows are ONE contiguous 12 KB block.  The staged instances of
// the kernel fetch that block with LDS-DMA (`global_load_lds_dwordx4`: 1 KB per wave-instruction, whole lines, no staging VGPRs),
// issued first thing -- in flight under the mean / scale / rotation loads and the projection -- and every thread then reads its
// row with ds_read_b128.  The LDS image is swizzled: slot s of row r holds float4 (s ^ f(r)) of that row, f(r) = (r >> 1) & 3, so
// that eight lanes reading "float4 q of my row" (48-dword pitch) cover all 32 banks; LDS-DMA writes lane l to base + 16 l, so the
// swizzle is applied to the per-lane GLOBAL address (within one 64-byte quarter line: still whole lines).  Same coefficients in
// the same left-to-right summation as sh_dot: bit-identical colours.
constexpr int SH_STAGE_F4 = 12 * 64;   // float4 per staged wave (M = 16: 12 per row)
constexpr int SH_STAGE_THREADS = 832;  // 13 waves x 12 KB = 156 KB of the CU's 160 KB, the binning arrays beside them
template <int K>
__device__ __forceinline__ void sh_dot_staged(const float (&B)[16], const float4* row, int f, float& a0, float& a1, float& a2)
{
    constexpr int NQ = (3 * K + 3) / 4;
    float v[4 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 t = row[q ^ f];
        v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) a0 += B[k] * v[3 * k], a1 += B[k] * v[3 * k + 1], a2 += B[k] * v[3 * k + 2];
}

// ------------------------------------------------------------------------------------------------
// K1: one thread per Gaussian, plus the first step of the binning (hgs_common.h, BIN_*):
// BIN_BY_CELL (large frames): the workgroup takes part in a counting sort of the Gaussians by binning cell (the BIN_CELL x
//   BIN_CELL tiles that hold the first tile of the rectangle) -- an LDS histogram whose returning adds rank the workgroup's
//   Gaussians inside each cell, ONE returning atomic per touched cell on the global cell counters, and (cell, slot inside
//   the cell) left in cell_slot for the scatter kernel (binning.hip).  The binning groups of the count and emit kernels are
//   runs of that order: neighbours on screen, so a group touches a few dozen tiles, many times each, instead of every tile
//   once or twice.
// BIN_IN_ORDER (small frames, where two more launches cost more than they save): the workgroup IS a binning group of
//   consecutive Gaussians and counts its pairs per tile in an LDS histogram while the rectangles are in registers, takes
//   ONE returning atomic per touched tile on the global per-tile counters -- the value returned is where this group's run
//   starts inside the tile's segment -- and leaves it in run_start[group][tile] for emit.
// BIN_NONE: no binning work (very large tile counts: count_kernel / emit_kernel<false> on global atomics).
// A second set of Gaussians behind the first (hgs_segment): Gaussian i >= P1 is element i - P1 of these arrays.  The
// kernels pick the array set per thread -- a handful of selects -- and index it with the segment-local index.
struct SecondInputs {
    int P1, M;  // Gaussians in the first set (== P when there is no second one); SH coefficients stored per Gaussian of the second
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rots, *cov3D_precomp;
};
struct SecondGrads {
    float *dL_dopacity, *dL_dcolors, *dL_dmeans3D, *dL_dsh, *dL_dscale, *dL_drot, *dL_dcov3D;
};
// Gradients of the FIRST set's Gaussians from ANOTHER render of the same step (hgs_backward_args.add_*): the kernel adds them
// to its own before it stores -- the sum autograd would form with one elementwise kernel per tensor.  All NULL: none.
struct FirstAdds {
    const float *dL_dopacity, *dL_dcolors, *dL_dmeans3D, *dL_dsh, *dL_dscale, *dL_drot, *dL_dcov3D;
};
// (a plain store where nothing is added: v + 0.0f would turn a -0.0f into +0.0f)
__device__ __forceinline__ float plus(float v, const float* other, size_t idx) { return other ? v + other[idx] : v; }

// H16 (BIN_IN_ORDER on frames beyond BIN_LDS_TILES tiles): the per-tile LDS counters are 16-bit halves (binning_walk.h, TileHist)
template <int MODE, bool STAGE = false, bool H16 = false>  // blockDim.x = bin_group_for() (<= 1024, ~250 workgroups) unless BIN_NONE: 256; STAGE: at most SH_STAGE_THREADS
__global__ void __launch_bounds__(STAGE ? SH_STAGE_THREADS : MODE != BIN_NONE ? BIN_GROUP : 256)
preprocess_kernel(int P, Camera cam, const float* __restrict__ means3D_, const float* __restrict__ shs_,
                  const float* __restrict__ colors_precomp_, const float* __restrict__ opacities_,
                  const float* __restrict__ scales_, const float* __restrict__ rots_,
                  const float* __restrict__ cov3D_precomp_, SecondInputs in2, const float* __restrict__ V,
                  const float* __restrict__ F, const float* __restrict__ campos, Splat* __restrict__ splats,
                  uint32_t* __restrict__ tiles_touched, int32_t* __restrict__ radii, uint8_t* __restrict__ visible,
                  uint32_t* __restrict__ counters, uint2* __restrict__ cell_slot, uint32_t* __restrict__ run_start,
                  float4* __restrict__ zero_accum, int big_per_group)
{
    const int NT = MODE != BIN_NONE ? (int)blockDim.x : 256;
    // BIN_BY_CELL: [num_cells] population of each cell in this workgroup, [num_cells] its base; BIN_IN_ORDER: [num_tiles] pairs
    extern __shared__ uint32_t bin_lds[];
    const int num_tiles = cam.gx * cam.gy;
    const int cells_x = (cam.gx + BIN_CELL - 1) / BIN_CELL, num_cells = cells_x * ((cam.gy + BIN_CELL - 1) / BIN_CELL);
    const int i = blockIdx.x * NT + threadIdx.x;
    // ---- STAGE: the wave's SH rows on their way into LDS before anything else is asked of memory ----
    const int lane = threadIdx.x & 63;
    bool staged = false;        // (wave-uniform) this wave's rows are one block of one array: they come through LDS
    const float4* my_row = nullptr;
    if (STAGE) {
        const int wave_first = __builtin_amdgcn_readfirstlane(i - lane);
        const bool wave_second = wave_first >= in2.P1;
        const int seg_end = wave_second ? P : in2.P1;
        const int rows = min(64, seg_end - wave_first);
        const float* shs_w = wave_second ? in2.shs : shs_;
        staged = wave_first < P && shs_w != nullptr && (wave_second ? in2.M : cam.M) == 16 && cam.D > 0 && wave_first + rows >= min(wave_first + 64, P);
        const int bin_words = MODE == BIN_BY_CELL ? 2 * (num_cells + 1) : MODE == BIN_IN_ORDER ? num_tiles : 0;
        float4* stage = reinterpret_cast<float4*>(bin_lds + ((bin_words + 3) & ~3)) + (size_t)(threadIdx.x >> 6) * SH_STAGE_F4;
        my_row = stage + lane * 12;
        if (staged) {
            const int nq = (3 * (cam.D + 1) * (cam.D + 1) + 3) / 4;   // float4 per row that hold coefficients of the active degree
            const float4* src = reinterpret_cast<const float4*>(shs_w + (size_t)(wave_second ? wave_first - in2.P1 : wave_first) * 48);
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int slot = it * 64 + lane, r = slot / 12, sl = slot - r * 12, q = sl ^ ((r >> 1) & 3);
                if (r < rows && q < nq)
                    __builtin_amdgcn_global_load_lds(src + r * 12 + q, (__attribute__((address_space(3))) void*)(stage + it * 64), 16, 0, 0);
            }
        }
    }
    if (MODE != BIN_NONE)
        for (int c = threadIdx.x; c < (MODE == BIN_BY_CELL ? num_cells + 1 : H16 ? (num_tiles + 1) / 2 : num_tiles); c += NT) bin_lds[c] = 0;  // visible after the barrier below
    // Housekeeping that would otherwise be another launch: when the caller will run backward, its [P,12] gradient
    // accumulator is zeroed here, fully coalesced.
    if (zero_accum) {
        const size_t base = (size_t)blockIdx.x * (3 * NT), end = (size_t)P * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (base + k * NT + threadIdx.x < end) zero_accum[base + k * NT + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    Splat out;
    out.x = out.y = out.la = out.lb = out.lc = out.ca = out.cb = out.cc = out.log2_opacity = out.r = out.g = out.b = out.depth = 0.0f;
    out.radius = 0;
    out.clamped = 0;
    uint32_t touched = 0;
    int rect_minx = 0, rect_miny = 0, rect_width = 1;

    bool alive = false;
    if (i < P) {
        // which set of input arrays this Gaussian lives in, and its index there
        const bool second = i >= in2.P1;
        const size_t j = (size_t)(second ? i - in2.P1 : i);
        const float* means3D = second ? in2.means3D : means3D_;
        const float* shs = second ? in2.shs : shs_;
        const float* colors_precomp = second ? in2.colors_precomp : colors_precomp_;
        const float* opacities = second ? in2.opacities : opacities_;
        const float* scales = second ? in2.scales : scales_;
        const float* rots = second ? in2.rots : rots_;
        const float* cov3D_precomp = second ? in2.cov3D_precomp : cov3D_precomp_;
        const int M = second ? in2.M : cam.M;
        const float x = means3D[3 * j], y = means3D[3 * j + 1], z = means3D[3 * j + 2];
        float pv[3];
        pv[0] = V[0] * x + V[4] * y + V[8] * z + V[12];
        pv[1] = V[1] * x + V[5] * y + V[9] * z + V[13];
        pv[2] = V[2] * x + V[6] * y + V[10] * z + V[14];
        alive = pv[2] > NEAR_Z;  // NaN culls
        if (alive) {
            float hx = F[0] * x + F[4] * y + F[8] * z + F[12];
            float hy = F[1] * x + F[5] * y + F[9] * z + F[13];
            float hw = F[3] * x + F[7] * y + F[11] * z + F[15];
            float pw = 1.0f / (hw + 0.0000001f);
            float ndcx = hx * pw, ndcy = hy * pw;

            float S[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) S[k] = cov3D_precomp[6 * j + k];
            } else {
                cov3d_from_scale_rot(scales + 3 * j, cam.mod, rots + 4 * j, S);
            }
            Ewa e;
            ewa_project(pv, cam, V, S, e);
            float det = e.a * e.c - e.b * e.b;
            // (A.2 step 5 culls det == 0; a projected covariance that is not POSITIVE DEFINITE -- det < 0 or a < 0, reachable only
            //  through a non-PSD cov3D_precomp -- is culled here as well: the blend kernels' Cholesky form of the exponent exists
            //  only for a positive-definite conic.  Documented in include/hgs_rasterizer.h; the oracle mirrors the rule behind
            //  oracle_set_cull_non_pd.  NaN compares false.)
            alive = det > 0.0f && e.a > 0.0f;
            if (alive) {
                float det_inv = 1.0f / det;
                float mid = 0.5f * (e.a + e.c);
                float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                float l1 = mid + sq, l2 = mid - sq;
                float radf = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
                float px = ((ndcx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;
                float py = ((ndcy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
                float fminx = fminf((float)cam.gx, fmaxf(0.0f, (px - radf) / 16.0f));
                float fmaxx = fminf((float)cam.gx, fmaxf(0.0f, (px + radf + 15.0f) / 16.0f));
                float fminy = fminf((float)cam.gy, fmaxf(0.0f, (py - radf) / 16.0f));
                float fmaxy = fminf((float)cam.gy, fmaxf(0.0f, (py + radf + 15.0f) / 16.0f));
                // non-finite centre or radius: culled (NaN compares false)
                bool finite = (fabsf(px) <= 3.0e38f) && (fabsf(py) <= 3.0e38f) && (radf <= 1.0e9f);
                int minx = (int)fminx, maxx = (int)fmaxx, miny = (int)fminy, maxy = (int)fmaxy;
                alive = finite && maxx > minx && maxy > miny;
                if (alive) {
                    out.x = px;
                    out.y = py;
                    // stored for the log2 domain (hgs_common.h): the half-conic (-conic.x/2, -conic.y, -conic.z/2) -- an exact
                    // rescale -- times LOG2E (one rounding), and log2(opacity)
                    out.ca = (-0.5f * (e.c * det_inv)) * LOG2E;
                    out.cb = (-(-e.b * det_inv)) * LOG2E;
                    out.cc = (-0.5f * (e.a * det_inv)) * LOG2E;
                    // ... and its Cholesky factors for the blend kernels (hgs_common.h): with A' = -ca etc.,
                    // C' - lb^2 = (A'C' - B'^2/4) / A' = (LOG2E/2)^2 / (det A') = (LOG2E/2) / cov2D.yy -- no cancellation
                    out.la = sqrtf(-out.ca);
                    out.lb = (-0.5f * out.cb) / out.la;
                    out.lc = sqrtf((0.5f * LOG2E) / e.c);
                    out.log2_opacity = __log2f(opacities[j]);
                    out.depth = pv[2];
                    out.radius = (int32_t)radf;
                    touched = (uint32_t)((maxx - minx) * (maxy - miny));
                    rect_minx = minx, rect_miny = miny, rect_width = maxx - minx;
                    if (shs) {
                        float dx = x - campos[0], dy = y - campos[1], dz = z - campos[2];
                        float len = sqrtf(dx * dx + dy * dy + dz * dz);
                        dx = dx / len, dy = dy / len, dz = dz / len;
                        float B[16];
                        sh_basis(cam.D, dx, dy, dz, B);
                        const float* sh = shs + j * M * 3;
                        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
                        if (STAGE && staged) {
                            // the wave's LDS-DMA has landed once its own vmcnt says so (nothing else orders a ds_read behind it)
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const int f = (lane >> 1) & 3;
                            switch (cam.D) {
                                case 1: sh_dot_staged<4>(B, my_row, f, acc0, acc1, acc2); break;
                                case 2: sh_dot_staged<9>(B, my_row, f, acc0, acc1, acc2); break;
                                default: sh_dot_staged<16>(B, my_row, f, acc0, acc1, acc2); break;
                            }
                        } else
                        // the number of coefficients is a compile-time constant inside each case, so all their loads are
                        // issued before the first is waited for (a runtime trip count made it one round trip per coefficient)
                        switch (cam.D) {
                            case 0: sh_dot<1>(B, sh, acc0, acc1, acc2); break;
                            case 1: sh_dot<4>(B, sh, acc0, acc1, acc2); break;
                            case 2: sh_dot<9>(B, sh, acc0, acc1, acc2); break;
                            default: sh_dot<16>(B, sh, acc0, acc1, acc2); break;
                        }
                        acc0 += 0.5f, acc1 += 0.5f, acc2 += 0.5f;
                        out.clamped = (acc0 < 0.0f ? 1u : 0u) | (acc1 < 0.0f ? 2u : 0u) | (acc2 < 0.0f ? 4u : 0u);
                        out.r = fmaxf(acc0, 0.0f), out.g = fmaxf(acc1, 0.0f), out.b = fmaxf(acc2, 0.0f);
                    } else {
                        out.r = colors_precomp[3 * j], out.g = colors_precomp[3 * j + 1];
                        out.b = colors_precomp[3 * j + 2];
                    }
                }
            }
        }
        if (!alive) {
            out.x = out.y = out.la = out.lb = out.lc = out.ca = out.cb = out.cc = out.log2_opacity = out.depth = 0.0f;
            out.radius = 0;
            touched = 0;
        }
        float4* dst = reinterpret_cast<float4*>(splats + i);
        dst[0] = make_float4(out.x, out.y, out.la, out.lb);
        dst[1] = make_float4(out.lc, out.log2_opacity, out.r, out.g);
        dst[2] = make_float4(out.b, out.depth, __int_as_float(out.radius), __uint_as_float(out.clamped));
        dst[3] = make_float4(out.ca, out.cb, out.cc, out.log2_opacity);
        tiles_touched[i] = touched;
        radii[i] = out.radius;
        if (visible) visible[i] = out.radius > 0 ? 1 : 0;  // the renderer's `radii > 0`, without its kernel
    }

    if (MODE == BIN_BY_CELL) {
        uint32_t* population = bin_lds;              // [num_cells + 1]: the last entry is the BIG cell
        uint32_t* base = bin_lds + num_cells + 1;
        // the cell of the rectangle's first tile -- but a BIG splat (more than BIN_SPREAD_MIN tiles) goes to a cell of its own kind:
        // every splat that reaches the image's top-left corner has its first tile at (0, 0), and a trained scene keeps a thousand
        // background splats a hundred and more pixels across (some cover the whole frame); together in cell 0 they made ONE
        // binning group of 96 000 pairs where the mean is 6 700 (round 3).  Round 4 dealt them to pseudo-random cells: balanced,
        // but three or four in EVERY group, whose tile window -- what the count and emit kernels pay per tile of -- then was the
        // whole screen.  Round 5: the big cell, scattered behind the others into groups of big_per_group (cell_scatter_kernel);
        // big_per_group == 0 keeps round 4's spreading.
        const int cell = !touched ? -1
                         : touched > BIN_SPREAD_MIN ? (big_per_group > 0 ? num_cells : (int)((((uint32_t)i * 2654435761u) >> 12) % (uint32_t)num_cells))
                                                    : (rect_miny / BIN_CELL) * cells_x + rect_minx / BIN_CELL;
        __syncthreads();  // population zeroed
        const uint32_t rank = cell >= 0 ? atomicAdd(&population[cell], 1u) : 0u;
        __syncthreads();
        for (int c = threadIdx.x; c <= num_cells; c += NT) {
            const uint32_t n = population[c];
            if (n) base[c] = atomicAdd(&counters[c], n);
        }
        __syncthreads();
        if (i < P) cell_slot[i] = cell >= 0 ? make_uint2((uint32_t)cell, base[cell] + rank) : make_uint2(0xFFFFFFFFu, 0u);
    }
    if (MODE == BIN_IN_ORDER) {
        const TileHist<H16> hist{bin_lds};
        SplatRect mine;
        mine.x = mine.y = 0.f, mine.A = mine.C = -1.f, mine.B = 0.f, mine.thr = 0.f, mine.depth_bits = 0;
        mine.minx = rect_minx, mine.miny = rect_miny, mine.width = rect_width, mine.cnt = touched;
        __syncthreads();  // hist zeroed
        for_each_pair(mine, [&](int, int tx, int ty, const SplatRect&) { hist.add(ty * cam.gx + tx); });
        __syncthreads();
        // sixteen tiles per thread and round: the returning atomics of a round are all in flight together (the SMPL template's
        // groups are ONE wave each: the 1 024 tiles of a 512x512 frame are one round trip, not two)
        constexpr int U = 16;
        uint32_t* my_runs = run_start + (size_t)blockIdx.x * num_tiles;
        for (int t0 = threadIdx.x; t0 < num_tiles; t0 += U * NT) {
            uint32_t c[U], base[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * NT;
                c[u] = t < num_tiles ? hist.get(t) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) base[u] = c[u] ? atomicAdd(&counters[t0 + u * NT], c[u]) : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c[u]) my_runs[t0 + u * NT] = base[u];
        }
    }
}

void launch_preprocess(const hgs_forward_args& a, const Camera& cam, Splat* splats, uint32_t* tiles_touched, int mode,
                       uint32_t* counters, uint2* cell_slot, uint32_t* run_start, int group, int big_per_group, hipStream_t st)
{
    const int P = a.P + a.seg2.P;
    const SecondInputs in2{a.P, a.seg2.M, a.seg2.means3D, a.seg2.shs, a.seg2.colors_precomp, a.seg2.opacities, a.seg2.scales,
                           a.seg2.rotations, a.seg2.cov3D_precomp};
#define HGS_K1_ARGS P, cam, a.means3D, a.shs, a.colors_precomp, a.opacities, a.scales, a.rotations, a.cov3D_precomp, in2, a.s.viewmatrix, \
                    a.s.projmatrix, a.s.campos, splats, tiles_touched, a.radii, a.visible
    // SH rows through LDS (degree >= 1 on the [P,16,3] layout): 12 KB per wave next to the binning arrays, when a binning group's
    // waves fit one CU's LDS.  OFF by default (HGS_K1_STAGE_SH=1 turns it on): round 5 measured it no faster -- C2 21.7-22.0 us
    // against 21.4-21.5 (HIP events), 100k 17.1 against 16.9, 50k 14.0 against 13.3 -- the kernel is a chain of dependent round trips
    // (mean -> scale / rotation -> opacity / SH -> cell or tile atomics), not short of load bandwidth, and the staging buffers cut the
    // CU from 20 resident waves to 13.  Bit-identical either way (tests/test_gpu_parity.py::test_sh_rows_through_lds...).
    const bool wants_stage = switches().k1_stage_sh && a.shs && a.s.sh_degree > 0 && (a.M == 16 || (a.seg2.P > 0 && a.seg2.M == 16));
    const size_t stage_bytes = sizeof(float4) * SH_STAGE_F4;
    if (mode == BIN_BY_CELL) {
        const size_t bin_bytes = 2 * sizeof(uint32_t) * (num_cells_of(cam.gx, cam.gy) + 1);
        // (the staging buffers cut the CU from 20 resident waves to 13: the launch stays ONE round of workgroups only when a whole
        //  binning group's waves fit one CU's LDS -- 256-thread workgroups, three per CU, were measured at 36.5 us against 21.4 on C2:
        //  782 workgroups on 768 slots are two rounds)
        static const bool big_lds_ok = hipFuncSetAttribute((const void*)preprocess_kernel<BIN_BY_CELL, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           160 * 1024) == hipSuccess;
        const size_t staged_total = ((bin_bytes + 15) & ~(size_t)15) + (size_t)((group + 63) / 64) * stage_bytes;
        if (wants_stage && big_lds_ok && group <= 832 && staged_total <= 160 * 1024) {
            hipLaunchKernelGGL((preprocess_kernel<BIN_BY_CELL, true>), dim3((P + group - 1) / group), dim3(group), staged_total, st,
                               HGS_K1_ARGS, counters, cell_slot, nullptr, (float4*)a.grad_accum_to_zero, big_per_group);
        } else
            hipLaunchKernelGGL(preprocess_kernel<BIN_BY_CELL>, dim3((P + group - 1) / group), dim3(group), bin_bytes, st, HGS_K1_ARGS, counters, cell_slot, nullptr,
                               (float4*)a.grad_accum_to_zero, big_per_group);
    } else if (mode == BIN_IN_ORDER) {
        const size_t bin_bytes = sizeof(uint32_t) * cam.gx * cam.gy;
        const size_t staged_total = ((bin_bytes + 15) & ~(size_t)15) + (size_t)((group + 63) / 64) * stage_bytes;
        // (more than 64 KB of dynamic LDS has to be allowed once per kernel)
        static const bool big_lds_ok = hipFuncSetAttribute((const void*)preprocess_kernel<BIN_IN_ORDER, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           160 * 1024) == hipSuccess;
        if (cam.gx * cam.gy > BIN_LDS_TILES) {   // (16-bit counters; the SH staging is not combined with it)
            if (hipFuncSetAttribute((const void*)preprocess_kernel<BIN_IN_ORDER, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)TileHist<true>::bytes(cam.gx * cam.gy)) != hipSuccess)
                (void)hipGetLastError();
            hipLaunchKernelGGL((preprocess_kernel<BIN_IN_ORDER, false, true>), dim3((P + group - 1) / group), dim3(group), TileHist<true>::bytes(cam.gx * cam.gy), st,
                               HGS_K1_ARGS, counters, nullptr, run_start, (float4*)a.grad_accum_to_zero, big_per_group);
        } else if (wants_stage && group <= SH_STAGE_THREADS && staged_total <= (big_lds_ok ? 160 * 1024 : 64 * 1024))
            hipLaunchKernelGGL((preprocess_kernel<BIN_IN_ORDER, true>), dim3((P + group - 1) / group), dim3(group), staged_total, st, HGS_K1_ARGS, counters, nullptr,
                               run_start, (float4*)a.grad_accum_to_zero, big_per_group);
        else
            hipLaunchKernelGGL(preprocess_kernel<BIN_IN_ORDER>, dim3((P + group - 1) / group), dim3(group), bin_bytes, st, HGS_K1_ARGS, counters, nullptr, run_start,
                               (float4*)a.grad_accum_to_zero, big_per_group);
    } else
        hipLaunchKernelGGL(preprocess_kernel<BIN_NONE>, dim3((P + 255) / 256), dim3(256), 0, st, HGS_K1_ARGS, nullptr, nullptr,
                           nullptr, (float4*)a.grad_accum_to_zero, big_per_group);
#undef HGS_K1_ARGS
}

// ------------------------------------------------------------------------------------------------
// K8+K9 fused: one thread per Gaussian, only radius > 0 does work; every output element is written here (zeros for
// culled Gaussians and SH coefficients above the active degree), the caller pre-zeroes nothing.
//
// The SH rows -- 192 bytes per Gaussian on the [P,16,3] layout the HUGS models keep -- are moved by the WAVE, not by the
// thread: a thread reading or writing its own row makes every load / store instruction touch 64 different cache lines,
// one dword each, and the kernel ran on the address path instead of on HBM (WRITE_SIZE 1.32x the algorithmic bytes).
// Here a wave's 64 rows are one contiguous 12 KB block: it is written (and, with coop_mode bit 1, read) with float4
// accesses that are whole lines, staged through LDS -- a row per lane with an odd stride of 3 K + 1 dwords (K = coefficients of the
// active degree; conflict-free) -- and the coefficients above the active degree (180 of every 192 bytes at degree 0)
// are written as zeros straight from registers.  Waves whose rows are not one block of one array (M != 16, a wave that
// straddles the two segments) keep the per-thread path.
constexpr int SH_ROW_F4 = 12;  // float4 per row at M = 16
#ifndef HGS_K8_WAVES   // (A/B builds: tools/ab_build.sh name "-DHGS_K8_WAVES=5 -DHGS_K8_SH_CHUNK=8")
#define HGS_K8_WAVES 4
#endif
template <bool ADD>   // ADD: the first set's gradients are added to another render's (FirstAdds)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HGS_K8_WAVES)))   // (4: <= 128 VGPRs; 5: <= 96)
preprocess_backward_kernel(int P, Camera cam, const float* __restrict__ means3D_, const float* __restrict__ shs_,
                           const float* __restrict__ opacities_, const float* __restrict__ scales_, const float* __restrict__ rots_,
                           const float* __restrict__ cov3D_precomp_, SecondInputs in2, const float* __restrict__ V,
                           const float* __restrict__ F, const float* __restrict__ campos,
                           const Splat* __restrict__ splats, const float* __restrict__ grad_accum,
                           float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity_,
                           float* __restrict__ dL_dcolors_, float* __restrict__ dL_dmeans3D_,
                           float* __restrict__ dL_dsh_, float* __restrict__ dL_dscale_, float* __restrict__ dL_drot_,
                           float* __restrict__ dL_dcov3D_, SecondGrads out2, FirstAdds add1, int coop_mode)
{
    extern __shared__ float sh_stage[];  // [waves][64][3 K + 1]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // ---- wave-level view of the SH rows (all values wave-uniform) ----
    const int wave_first = __builtin_amdgcn_readfirstlane(i - lane);   // joint index of lane 0's Gaussian
    if (wave_first >= P) return;
    const int Kact = (cam.D + 1) * (cam.D + 1), row_stride = 3 * Kact + 1;
    const bool wave_second = wave_first >= in2.P1;
    const int seg_end = wave_second ? P : in2.P1;                       // one past the last Gaussian of the wave's segment
    const int rows = min(64, seg_end - wave_first);                     // rows of the wave that lie in its segment
    // coop_mode: bit 0 = the wave stores the dL/dsh rows, bit 1 = it also loads the SH rows (else every thread its own)
    const bool coop = (coop_mode & 1) && shs_ != nullptr && (wave_second ? in2.M : cam.M) == 16 && wave_first + rows >= min(wave_first + 64, P);
    const bool coop_load = coop && (coop_mode & 2) && cam.D > 0;
    float* stage = sh_stage + (size_t)(threadIdx.x >> 6) * 64 * row_stride;
    const size_t wave_row0 = (size_t)(wave_second ? wave_first - in2.P1 : wave_first);
    if (coop_load) {
        // the wave's rows, whole lines at a time, into LDS (only the coefficients of the active degree are kept)
        const float4* src = reinterpret_cast<const float4*>((wave_second ? in2.shs : shs_) + wave_row0 * 48);
#pragma unroll
        for (int it = 0; it < SH_ROW_F4; ++it) {
            const int idx = it * 64 + lane, row = idx / SH_ROW_F4, e = (idx - row * SH_ROW_F4) * 4;
            if (row < rows && e < 3 * Kact) {
                const float4 v = src[idx];
                float* d = stage + row * row_stride + e;
                d[0] = v.x;
                if (e + 1 < 3 * Kact) d[1] = v.y;
                if (e + 2 < 3 * Kact) d[2] = v.z;
                if (e + 3 < 3 * Kact) d[3] = v.w;
            }
        }
    }
    // (from here on a thread without a Gaussian only takes part in the wave's row stores)
    const bool has = i < P;
    if (!has) i = P - 1;
    // the set of arrays (inputs and gradients) this Gaussian lives in, and its index there; the accumulator, the splat
    // record and dL/dmean2D are indexed by the joint index i
    const bool second = i >= in2.P1;
    const size_t j = (size_t)(second ? i - in2.P1 : i);
    const float* means3D = second ? in2.means3D : means3D_;
    const float* shs = second ? in2.shs : shs_;
    const float* opacities = second ? in2.opacities : opacities_;
    const float* scales = second ? in2.scales : scales_;
    const float* rots = second ? in2.rots : rots_;
    const float* cov3D_precomp = second ? in2.cov3D_precomp : cov3D_precomp_;
    float* dL_dopacity = second ? out2.dL_dopacity : dL_dopacity_;
    float* dL_dcolors = second ? out2.dL_dcolors : dL_dcolors_;
    float* dL_dmeans3D = second ? out2.dL_dmeans3D : dL_dmeans3D_;
    float* dL_dsh = second ? out2.dL_dsh : dL_dsh_;
    float* dL_dscale = second ? out2.dL_dscale : dL_dscale_;
    float* dL_drot = second ? out2.dL_drot : dL_drot_;
    float* dL_dcov3D = second ? out2.dL_dcov3D : dL_dcov3D_;
    const int M = second ? in2.M : cam.M;
    // the other render's gradients of this Gaussian (first set only; NULL: nothing to add)
    const float* ad_opacity = !ADD || second ? nullptr : add1.dL_dopacity;
    const float* ad_colors = !ADD || second ? nullptr : add1.dL_dcolors;
    const float* ad_means3D = !ADD || second ? nullptr : add1.dL_dmeans3D;
    const float* ad_sh = !ADD || second ? nullptr : add1.dL_dsh;
    const float* ad_scale = !ADD || second ? nullptr : add1.dL_dscale;
    const float* ad_rot = !ADD || second ? nullptr : add1.dL_drot;
    const float* ad_cov3D = !ADD || second ? nullptr : add1.dL_dcov3D;
    // accumulator record written by the blend-backward atomics, raw moments of u = G dL/dalpha over the pixels:
    //   sum u dx, sum u dy, sum u dx^2, sum u dx dy | sum u dy^2, sum u, dL/dr, dL/dg | dL/db - - -
    // with u = opacity G dL/dalpha (the uncapped alpha times dL/dalpha).  Turned here, once per Gaussian, into
    // dL/dmean2D (pixel units -> NDC-scaled, A.6 quirk 3), dL/dconic and dL/dopacity with the Gaussian's own opacity
    // and half-conic (A, B, C) = -(conic.x/2, conic.y, conic.z/2) (the record holds LOG2E times it):
    //   dG/ddx = G (2A dx + B dy),  dG/dconic.x = -G dx^2 / 2,  dalpha/dopacity = G.
    float4 acc0 = reinterpret_cast<const float4*>(grad_accum)[3 * (size_t)i];
    float4 acc1 = reinterpret_cast<const float4*>(grad_accum)[3 * (size_t)i + 1];
    const float acc_b = grad_accum[12 * (size_t)i + 8];
    // (round 4) the Gaussian's small inputs are fetched here, with the accumulator and the record and whether or not it turns
    // out to be visible: one memory round trip instead of three dependent ones (record -> opacity -> mean / scale / rotation)
    // in a kernel whose whole grid is resident at once, three waves per SIMD, and lasts as long as one wave's chain of loads:
    // C2 28.4 -> 27.1 us, C4 22.4 -> 20.8, the 110k human 12.6 -> 11.3, the SMPL template 8.0 -> 6.9 (same box).
    // (The forward kernel did NOT gain from the same change -- C2 18.0 us either way, the 110k human 13.9 -> 16.0 -- nor from
    //  starting its cell-counter atomics before the SH row fetches so that the two round trips overlap: C2 18.1 -> 21.2 us.)
    const float in_x = means3D[3 * j], in_y = means3D[3 * j + 1], in_z = means3D[3 * j + 2], in_op = opacities[j];
    float in_sc[3] = {0.f, 0.f, 0.f}, in_q[4] = {0.f, 0.f, 0.f, 0.f}, in_S[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) in_S[k] = cov3D_precomp[6 * j + k];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) in_sc[k] = scales[3 * j + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) in_q[k] = rots[4 * j + k];   // (scalar loads: the caller's tensor need not be 16-byte aligned)
    }
    {
        const float4 hc = reinterpret_cast<const float4*>(splats + i)[3];   // the half-conic quarter of the record
        const bool live = __float_as_int(reinterpret_cast<const float4*>(splats + i)[2].z) > 0;  // else: record unset
        const float A = live ? hc.x * LN2 : 0.0f, B = live ? hc.y * LN2 : 0.0f, C = live ? hc.z * LN2 : 0.0f;
        const float op = live ? in_op : 0.0f;
        const float sx = acc0.x, sy = acc0.y;
        acc0.x = (0.5f * (float)cam.W) * (2.0f * A * sx + B * sy);
        acc0.y = (0.5f * (float)cam.H) * (2.0f * C * sy + B * sx);
        acc0.z = -0.5f * acc0.z, acc0.w = -0.5f * acc0.w, acc1.x = -0.5f * acc1.x;
        acc1.y = op > 0.0f ? acc1.y / op : 0.0f;  // sum of G dL/dalpha
    }
    if (has) {
        dL_dmean2D[3 * (size_t)i] = acc0.x, dL_dmean2D[3 * (size_t)i + 1] = acc0.y, dL_dmean2D[3 * (size_t)i + 2] = 0.0f;
        dL_dopacity[j] = plus(acc1.y, ad_opacity, j);
        dL_dcolors[3 * j] = plus(acc1.z, ad_colors, 3 * j), dL_dcolors[3 * j + 1] = plus(acc1.w, ad_colors, 3 * j + 1);
        dL_dcolors[3 * j + 2] = plus(acc_b, ad_colors, 3 * j + 2);
    }

    const float4 tail = reinterpret_cast<const float4*>(splats + i)[2];
    const bool live = has && __float_as_int(tail.z) > 0;
    float* my_row = stage + lane * row_stride;  // this Gaussian's row of the wave's LDS stage (coop)
    if (has && !live) {
        // every output is fully written by this kernel (the caller does not pre-zero them)
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * j + k] = plus(0.0f, ad_means3D, 3 * j + k), dL_dscale[3 * j + k] = plus(0.0f, ad_scale, 3 * j + k);
        reinterpret_cast<float4*>(dL_drot)[j] = ad_rot ? reinterpret_cast<const float4*>(ad_rot)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * j + k] = plus(0.0f, ad_cov3D, 6 * j + k);
        if (shs && !coop)
            for (int k = 0; k < 3 * M; ++k) dL_dsh[j * M * 3 + k] = k < 3 * Kact ? plus(0.0f, ad_sh, j * M * 3 + k) : 0.0f;
    }
    if (coop && !live)
        for (int k = 0; k < 3 * Kact; ++k) my_row[k] = plus(0.0f, ad_sh, j * M * 3 + k);
    // (no early return: every lane takes part in the wave's row stores at the end)
    auto per_gaussian = [&]() {
    const uint32_t clamped = __float_as_uint(tail.w);

    const float x = in_x, y = in_y, z = in_z;
    // SH backward first (round 6): its 48 coefficients, 16 basis values and 16 dot products are dead before the covariance chain's
    // intermediates come alive (the register allocation did not change for it: the chunked fetch above is what moves it)
    float sh_dm0 = 0.0f, sh_dm1 = 0.0f, sh_dm2 = 0.0f;
    if (shs) {
        float dr0 = (clamped & 1u) ? 0.0f : acc1.z;
        float dr1 = (clamped & 2u) ? 0.0f : acc1.w;
        float dr2 = (clamped & 4u) ? 0.0f : acc_b;
        float vx = x - campos[0], vy = y - campos[1], vz = z - campos[2];
        float len = sqrtf(vx * vx + vy * vy + vz * vz);
        float X = vx / len, Y = vy / len, Z = vz / len;
        const int D = cam.D;
        float B[16];
        sh_basis(D, X, Y, Z, B);
        const int K = (D + 1) * (D + 1);
        // shw[k] = sh[k] . dL/dcolour for the direction gradient, and dL/dsh[k] = B[k] dL/dcolour: the coefficient count
        // is a compile-time constant inside each case, so all loads are in flight together (see sh_dot)
        float shw[16];
        if (coop) {
            // rows through the wave's LDS stage: read from it (degree 0: the three floats straight from memory), written to it
            const float* sh_in = coop_load ? my_row : shs + j * M * 3;
            switch (D) {
                case 0: sh_backward_rows<1>(B, shs + j * M * 3, my_row, dr0, dr1, dr2, shw); break;
                case 1: sh_backward_rows<4>(B, sh_in, my_row, dr0, dr1, dr2, shw); break;
                case 2: sh_backward_rows<9>(B, sh_in, my_row, dr0, dr1, dr2, shw); break;
                default: sh_backward_rows<16>(B, sh_in, my_row, dr0, dr1, dr2, shw); break;
            }
        } else {
            const float* sh = shs + j * M * 3;
            float* dsh = dL_dsh + j * M * 3;
            switch (D) {
                case 0: sh_backward_rows<1>(B, sh, dsh, dr0, dr1, dr2, shw); break;
                case 1: sh_backward_rows<4>(B, sh, dsh, dr0, dr1, dr2, shw); break;
                case 2: sh_backward_rows<9>(B, sh, dsh, dr0, dr1, dr2, shw); break;
                default: sh_backward_rows<16>(B, sh, dsh, dr0, dr1, dr2, shw); break;
            }
            for (int k = 3 * K; k < 3 * M; ++k) dsh[k] = 0.0f;  // coefficients above the active degree
        }
        if (ad_sh) {   // (the other render ran at the same degree: above it both gradients are zero)
            float* row = coop ? my_row : dL_dsh + j * M * 3;
            for (int k = 0; k < 3 * K; ++k) row[k] += ad_sh[j * M * 3 + k];
        }
        float ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
#define SHW(k) shw[k]
        if (D > 0) {
            const float c1 = (float)0.4886025119029199;
            ddy += -c1 * SHW(1);
            ddz += c1 * SHW(2);
            ddx += -c1 * SHW(3);
            if (D > 1) {
                const float c20 = (float)1.0925484305920792, c21 = (float)-1.0925484305920792,
                            c22 = (float)0.31539156525252005, c23 = (float)-1.0925484305920792,
                            c24 = (float)0.5462742152960396;
                float xx = X * X, yy = Y * Y, zz = Z * Z;
                float w4 = SHW(4), w5 = SHW(5), w6 = SHW(6), w7 = SHW(7), w8 = SHW(8);
                ddx += c20 * Y * w4 + c22 * -2.0f * X * w6 + c23 * Z * w7 + c24 * 2.0f * X * w8;
                ddy += c20 * X * w4 + c21 * Z * w5 + c22 * -2.0f * Y * w6 + c24 * -2.0f * Y * w8;
                ddz += c21 * Y * w5 + c22 * 4.0f * Z * w6 + c23 * X * w7;
                if (D > 2) {
                    const float c30 = (float)-0.5900435899266435, c31 = (float)2.890611442640554,
                                c32 = (float)-0.4570457994644658, c33 = (float)0.3731763325901154,
                                c34 = (float)-0.4570457994644658, c35 = (float)1.445305721320277,
                                c36 = (float)-0.5900435899266435;
                    float w9 = SHW(9), w10 = SHW(10), w11 = SHW(11), w12 = SHW(12), w13 = SHW(13), w14 = SHW(14),
                          w15 = SHW(15);
                    ddx += c30 * 6.0f * X * Y * w9 + c31 * Y * Z * w10 + c32 * -2.0f * X * Y * w11 +
                           c33 * -6.0f * X * Z * w12 + c34 * (4.0f * zz - 3.0f * xx - yy) * w13 +
                           c35 * 2.0f * X * Z * w14 + c36 * (3.0f * xx - 3.0f * yy) * w15;
                    ddy += c30 * (3.0f * xx - 3.0f * yy) * w9 + c31 * X * Z * w10 +
                           c32 * (4.0f * zz - xx - 3.0f * yy) * w11 + c33 * -6.0f * Y * Z * w12 +
                           c34 * -2.0f * X * Y * w13 + c35 * -2.0f * Y * Z * w14 + c36 * -6.0f * X * Y * w15;
                    ddz += c31 * X * Y * w10 + c32 * 8.0f * Y * Z * w11 +
                           c33 * (6.0f * zz - 3.0f * xx - 3.0f * yy) * w12 + c34 * 8.0f * X * Z * w13 +
                           c35 * (xx - yy) * w14;
                }
            }
        }
#undef SHW
        float s2 = vx * vx + vy * vy + vz * vz;
        float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
        sh_dm0 = ((s2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * inv32;
        sh_dm1 = (-vx * vy * ddx + (s2 - vy * vy) * ddy - vz * vy * ddz) * inv32;
        sh_dm2 = (-vx * vz * ddx - vy * vz * ddy + (s2 - vz * vz) * ddz) * inv32;
    }
    float pv[3];
    pv[0] = V[0] * x + V[4] * y + V[8] * z + V[12];
    pv[1] = V[1] * x + V[5] * y + V[9] * z + V[13];
    pv[2] = V[2] * x + V[6] * y + V[10] * z + V[14];
    float S[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = in_S[k];
    } else {
        cov3d_from_scale_rot(in_sc, cam.mod, in_q, S);
    }
    Ewa e;
    ewa_project(pv, cam, V, S, e);
    const float a = e.a, b = e.b, c = e.c;
    const float gxx = acc0.z, gxy = acc0.w, gyy = acc1.x;
    const float denom = a * c - b * b;
    const float d2inv = 1.0f / (denom * denom + 0.0000001f);
    float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
    float dS[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (d2inv != 0.0f) {
        dL_da = d2inv * (-c * c * gxx + 2.0f * b * c * gxy + (denom - a * c) * gyy);
        dL_dc = d2inv * (-a * a * gyy + 2.0f * a * b * gxy + (denom - a * c) * gxx);
        dL_db = d2inv * 2.0f * (b * c * gxx - (denom + 2.0f * b * b) * gxy + a * b * gyy);
        dS[0] = e.T00 * e.T00 * dL_da + e.T00 * e.T10 * dL_db + e.T10 * e.T10 * dL_dc;
        dS[3] = e.T01 * e.T01 * dL_da + e.T01 * e.T11 * dL_db + e.T11 * e.T11 * dL_dc;
        dS[5] = e.T02 * e.T02 * dL_da + e.T02 * e.T12 * dL_db + e.T12 * e.T12 * dL_dc;
        dS[1] = 2.0f * e.T00 * e.T01 * dL_da + (e.T00 * e.T11 + e.T01 * e.T10) * dL_db + 2.0f * e.T10 * e.T11 * dL_dc;
        dS[2] = 2.0f * e.T00 * e.T02 * dL_da + (e.T00 * e.T12 + e.T02 * e.T10) * dL_db + 2.0f * e.T10 * e.T12 * dL_dc;
        dS[4] = 2.0f * e.T02 * e.T01 * dL_da + (e.T01 * e.T12 + e.T02 * e.T11) * dL_db + 2.0f * e.T11 * e.T12 * dL_dc;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * j + k] = plus(dS[k], ad_cov3D, 6 * j + k);

    // dL/dT (2x3) -> dL/dJ -> dL/dt (view-space mean), with the frustum-clamp masks (A.6 quirk 2)
    float u00 = S[0] * e.T00 + S[1] * e.T01 + S[2] * e.T02;
    float u01 = S[1] * e.T00 + S[3] * e.T01 + S[4] * e.T02;
    float u02 = S[2] * e.T00 + S[4] * e.T01 + S[5] * e.T02;
    float u10 = S[0] * e.T10 + S[1] * e.T11 + S[2] * e.T12;
    float u11 = S[1] * e.T10 + S[3] * e.T11 + S[4] * e.T12;
    float u12 = S[2] * e.T10 + S[4] * e.T11 + S[5] * e.T12;
    float dT00 = 2.0f * u00 * dL_da + u10 * dL_db, dT01 = 2.0f * u01 * dL_da + u11 * dL_db;
    float dT02 = 2.0f * u02 * dL_da + u12 * dL_db;
    float dT10 = 2.0f * u10 * dL_dc + u00 * dL_db, dT11 = 2.0f * u11 * dL_dc + u01 * dL_db;
    float dT12 = 2.0f * u12 * dL_dc + u02 * dL_db;
    float dJ00 = V[0] * dT00 + V[4] * dT01 + V[8] * dT02;
    float dJ02 = V[2] * dT00 + V[6] * dT01 + V[10] * dT02;
    float dJ11 = V[1] * dT10 + V[5] * dT11 + V[9] * dT12;
    float dJ12 = V[2] * dT10 + V[6] * dT11 + V[10] * dT12;
    float tz = 1.0f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = (e.x_in ? 1.0f : 0.0f) * (-cam.fx * tz2 * dJ02);
    float dty = (e.y_in ? 1.0f : 0.0f) * (-cam.fy * tz2 * dJ12);
    float dtz = -cam.fx * tz2 * dJ00 - cam.fy * tz2 * dJ11 + (2.0f * cam.fx * e.tx) * tz3 * dJ02 +
                (2.0f * cam.fy * e.ty) * tz3 * dJ12;
    float dm0 = V[0] * dtx + V[1] * dty + V[2] * dtz;
    float dm1 = V[4] * dtx + V[5] * dty + V[6] * dtz;
    float dm2 = V[8] * dtx + V[9] * dty + V[10] * dtz;

    // mean2D (NDC-scaled) -> mean
    {
        float hx = F[0] * x + F[4] * y + F[8] * z + F[12];
        float hy = F[1] * x + F[5] * y + F[9] * z + F[13];
        float hw = F[3] * x + F[7] * y + F[11] * z + F[15];
        float mw = 1.0f / (hw + 0.0000001f);
        float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        float g2x = acc0.x, g2y = acc0.y;
        dm0 += (F[0] * mw - F[3] * mul1) * g2x + (F[1] * mw - F[3] * mul2) * g2y;
        dm1 += (F[4] * mw - F[7] * mul1) * g2x + (F[5] * mw - F[7] * mul2) * g2y;
        dm2 += (F[8] * mw - F[11] * mul1) * g2x + (F[9] * mw - F[11] * mul2) * g2y;
    }

    // (the SH block ran first: its part of dL/dmean is added here, in the order it always was)
    dm0 += sh_dm0, dm1 += sh_dm1, dm2 += sh_dm2;
    dL_dmeans3D[3 * j] = plus(dm0, ad_means3D, 3 * j);
    dL_dmeans3D[3 * j + 1] = plus(dm1, ad_means3D, 3 * j + 1);
    dL_dmeans3D[3 * j + 2] = plus(dm2, ad_means3D, 3 * j + 2);

    // Sigma3D -> scale, quaternion (quaternion gradient w.r.t. the UN-normalised q)
    if (!cov3D_precomp) {
        const float mod = cam.mod;
        const float r = in_q[0], qx = in_q[1], qy = in_q[2], qz = in_q[3];
        const float s[3] = {mod * in_sc[0], mod * in_sc[1], mod * in_sc[2]};
        const float R[3][3] = {
            {1.0f - 2.0f * (qy * qy + qz * qz), 2.0f * (qx * qy - r * qz), 2.0f * (qx * qz + r * qy)},
            {2.0f * (qx * qy + r * qz), 1.0f - 2.0f * (qx * qx + qz * qz), 2.0f * (qy * qz - r * qx)},
            {2.0f * (qx * qz - r * qy), 2.0f * (qy * qz + r * qx), 1.0f - 2.0f * (qx * qx + qy * qy)}};
        const float Gs[3][3] = {{dS[0], 0.5f * dS[1], 0.5f * dS[2]},
                                {0.5f * dS[1], dS[3], 0.5f * dS[4]},
                                {0.5f * dS[2], 0.5f * dS[4], dS[5]}};
        float dR[3][3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            float dMc[3];
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) acc += Gs[ii][kk] * (R[kk][jj] * s[jj]);
                dMc[ii] = 2.0f * acc;
            }
            float ds = R[0][jj] * dMc[0] + R[1][jj] * dMc[1] + R[2][jj] * dMc[2];
            dL_dscale[3 * j + jj] = plus(cam.scale_grad_factor * ds, ad_scale, 3 * j + jj);
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) dR[ii][jj] = s[jj] * dMc[ii];
        }
        float4 dq;
        dq.x = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        dq.y = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.0f * qx * dR[1][1] - r * dR[1][2] +
                       qz * dR[2][0] + r * dR[2][1] - 2.0f * qx * dR[2][2]);
        dq.z = 2.0f * (-2.0f * qy * dR[0][0] + qx * dR[0][1] + r * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] -
                       r * dR[2][0] + qz * dR[2][1] - 2.0f * qy * dR[2][2]);
        dq.w = 2.0f * (-2.0f * qz * dR[0][0] - r * dR[0][1] + qx * dR[0][2] + r * dR[1][0] - 2.0f * qz * dR[1][1] +
                       qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        if (ad_rot) {
            const float4 o = reinterpret_cast<const float4*>(ad_rot)[j];
            dq.x += o.x, dq.y += o.y, dq.z += o.z, dq.w += o.w;
        }
        reinterpret_cast<float4*>(dL_drot)[j] = dq;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dscale[3 * j + k] = plus(0.0f, ad_scale, 3 * j + k);
        reinterpret_cast<float4*>(dL_drot)[j] = ad_rot ? reinterpret_cast<const float4*>(ad_rot)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    };  // per_gaussian
    if (live) per_gaussian();

    if (coop) {
        // the wave's 64 rows of dL/dsh leave as whole lines: the staged coefficients of the active degree, zeros above it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>((wave_second ? out2.dL_dsh : dL_dsh_) + wave_row0 * 48);
#pragma unroll
        for (int it = 0; it < SH_ROW_F4; ++it) {
            const int idx = it * 64 + lane, row = idx / SH_ROW_F4, e = (idx - row * SH_ROW_F4) * 4;
            if (row < rows) {
                const float* r = stage + row * row_stride + e;
                float4 v;
                v.x = e < 3 * Kact ? r[0] : 0.0f, v.y = e + 1 < 3 * Kact ? r[1] : 0.0f;
                v.z = e + 2 < 3 * Kact ? r[2] : 0.0f, v.w = e + 3 < 3 * Kact ? r[3] : 0.0f;
                dst[idx] = v;
            }
        }
    }
}

void launch_preprocess_backward(const hgs_backward_args& a, const Camera& cam, const Splat* splats, hipStream_t st)
{
    const hgs_forward_args& f = a.fwd;
    const int P = f.P + f.seg2.P;
    int blocks = (P + 255) / 256;
    const SecondInputs in2{f.P, f.seg2.M, f.seg2.means3D, f.seg2.shs, f.seg2.colors_precomp, f.seg2.opacities, f.seg2.scales,
                           f.seg2.rotations, f.seg2.cov3D_precomp};
    const SecondGrads out2{a.seg2_dL_dopacity, a.seg2_dL_dcolors, a.seg2_dL_dmeans3D, a.seg2_dL_dsh, a.seg2_dL_dscales,
                           a.seg2_dL_drotations, a.seg2_dL_dcov3D};
    const FirstAdds add1{a.add_dL_dopacity, a.add_dL_dcolors, a.add_dL_dmeans3D, a.add_dL_dsh, a.add_dL_dscales, a.add_dL_drotations,
                         a.add_dL_dcov3D};
    const int forced = switches().k8_coop;   // HGS_K8_COOP: measurement override
    // default: the wave STORES the rows; loading them through LDS as well was measured slower at every degree (C2, degree 3:
    // 37.0 us per-thread / 32.0 store only / 39.4 load + store; degree 1: 38.3 / 25.5 / 27.9 -- the round trip through LDS
    // sits in front of the whole per-Gaussian computation)
    const int coop_mode = forced >= 0 ? forced : 1;
    const size_t stage_bytes = (f.shs && coop_mode) ? (size_t)256 * (3 * (cam.D + 1) * (cam.D + 1) + 1) * sizeof(float) : 0;  // [4 waves][64 rows][3 K + 1]
#define HGS_K8_ARGS dim3(blocks), dim3(256), stage_bytes, st, P, cam, f.means3D, f.shs, f.opacities, f.scales, f.rotations, f.cov3D_precomp, in2,       \
                    f.s.viewmatrix, f.s.projmatrix, f.s.campos, splats, a.grad_accum, a.dL_dmeans2D, a.dL_dopacity, a.dL_dcolors, a.dL_dmeans3D,  \
                    a.dL_dsh, a.dL_dscales, a.dL_drotations, a.dL_dcov3D, out2, add1, coop_mode
    if (a.add_dL_dopacity) hipLaunchKernelGGL(preprocess_backward_kernel<true>, HGS_K8_ARGS);
    else hipLaunchKernelGGL(preprocess_backward_kernel<false>, HGS_K8_ARGS);
#undef HGS_K8_ARGS
}

// K10
__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ V,
                                    uint8_t* __restrict__ present)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float z = V[2] * means3D[3 * (size_t)i] + V[6] * means3D[3 * (size_t)i + 1] + V[10] * means3D[3 * (size_t)i + 2] + V[14];
    present[i] = z > NEAR_Z ? 1 : 0;
}

void launch_mark_visible(int P, const float* means3D, const float* V, uint8_t* present, hipStream_t st)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, V, present);
}

}  // namespace hgs
