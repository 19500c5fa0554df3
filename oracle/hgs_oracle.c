/*
 * hgs_oracle.c -- CPU restatement of the differentiable Gaussian-splat rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ml-hugs_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED at the rasterizer boundary: the reference's implementation of this path
 * is the third-party module graphdeco-inria/diff-gaussian-rasterization
 * (/root/reference/.gitmodules:1-3), an un-vendored, un-pinned, EMPTY submodule in the
 * reference checkout, and the reference ships no tests or golden vectors for it.  This
 * file restates the published algorithm (Kerbl et al., "3D Gaussian Splatting",
 * SIGGRAPH 2023 sec. 4-6 + appendix; Zwicker et al., "EWA Splatting" 2002) as specified in
 * SURVEY.md Appendix A, anchored on the reference's only call site
 * (/root/reference/hugs/renderer/gs_renderer.py:126-152).  The sub-steps that DO exist
 * in the reference as Python are pinned against it by tests/golden (see
 * tests/golden/make_golden.py):
 *   - SH basis / constants       /root/reference/hugs/utils/spherical_harmonics.py:30-47,61-113
 *   - Sigma3D = R S^2 R^T + pack /root/reference/hugs/utils/general.py:161-210
 *   - projection / camera dicts  /root/reference/hugs/utils/graphics.py:76-96,
 *                                /root/reference/hugs/datasets/utils.py:15-53,64-124
 *
 * Build twice: -DREAL=float (the fp32 checker, op-for-op the order the HIP kernels use for
 * every index-affecting quantity) and -DREAL=double (accuracy reference).  Compile with
 * -ffp-contract=off so that no a*b+c is fused behind our back.
 *
 * Conventions: matrices are row-vector convention, flat index m[4*r+c],
 * x' = m[0]x + m[4]y + m[8]z + m[12].  Quaternion (w,x,y,z), NOT normalised.
 * SH tensor [P, M, 3].  Tile 16x16.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int g_upstream_scale_grad = 0; /* see oracle_preprocess_backward */
static int g_cull_non_pd = 0;         /* see oracle_preprocess */
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define TILE 16
#define R_(x) ((real)(x))

static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rabs_(real a) { return a < 0 ? -a : a; }
static inline real rsqrt_(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real rexp_(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real rceil_(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }

/* SH constants: spherical_harmonics.py:30-47 */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

int oracle_real_size(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------------------------ */
/* A.2 step 3: Sigma3D from (scale, quat).  M = R*diag(s); Sigma = M M^T.                */
static void cov3d_from_scale_rot(const real *scale, real mod, const real *q, real *cov6)
{
    real s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R00 = R_(1) - R_(2) * (y * y + z * z), R01 = R_(2) * (x * y - r * z), R02 = R_(2) * (x * z + r * y);
    real R10 = R_(2) * (x * y + r * z), R11 = R_(1) - R_(2) * (x * x + z * z), R12 = R_(2) * (y * z - r * x);
    real R20 = R_(2) * (x * z - r * y), R21 = R_(2) * (y * z + r * x), R22 = R_(1) - R_(2) * (x * x + y * y);
    real M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
    real M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
    real M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
    cov6[0] = M00 * M00 + M01 * M01 + M02 * M02;
    cov6[1] = M00 * M10 + M01 * M11 + M02 * M12;
    cov6[2] = M00 * M20 + M01 * M21 + M02 * M22;
    cov6[3] = M10 * M10 + M11 * M11 + M12 * M12;
    cov6[4] = M10 * M20 + M11 * M21 + M12 * M22;
    cov6[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

/* exported for the unit test against general.py:build_scaling_rotation/strip_symmetric */
void oracle_cov3d(int P, const real *scales, real mod, const real *rots, real *cov6)
{
    for (int i = 0; i < P; ++i) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov6 + 6 * i);
}

/* A.3: SH basis evaluated at unit direction d, for degree D. basis[16]. */
static void sh_basis(int D, real x, real y, real z, real *B)
{
    B[0] = R_(SH_C0);
    if (D > 0) {
        B[1] = -R_(SH_C1) * y;
        B[2] = R_(SH_C1) * z;
        B[3] = -R_(SH_C1) * x;
        if (D > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = R_(SH_C2[0]) * xy;
            B[5] = R_(SH_C2[1]) * yz;
            B[6] = R_(SH_C2[2]) * (R_(2) * zz - xx - yy);
            B[7] = R_(SH_C2[3]) * xz;
            B[8] = R_(SH_C2[4]) * (xx - yy);
            if (D > 2) {
                B[9] = R_(SH_C3[0]) * y * (R_(3) * xx - yy);
                B[10] = R_(SH_C3[1]) * xy * z;
                B[11] = R_(SH_C3[2]) * y * (R_(4) * zz - xx - yy);
                B[12] = R_(SH_C3[3]) * z * (R_(2) * zz - R_(3) * xx - R_(3) * yy);
                B[13] = R_(SH_C3[4]) * x * (R_(4) * zz - xx - yy);
                B[14] = R_(SH_C3[5]) * z * (xx - yy);
                B[15] = R_(SH_C3[6]) * x * (xx - R_(3) * yy);
            }
        }
    }
}

/* exported for the unit test against spherical_harmonics.py:eval_sh (no +0.5, no clamp) */
void oracle_eval_sh(int P, int M, int D, const real *shs, const real *dirs, real *out)
{
    int K = (D + 1) * (D + 1);
    for (int i = 0; i < P; ++i) {
        real B[16];
        sh_basis(D, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], B);
        for (int c = 0; c < 3; ++c) {
            real acc = 0;
            for (int k = 0; k < K; ++k) acc += B[k] * shs[(size_t)i * M * 3 + 3 * k + c];
            out[3 * i + c] = acc;
        }
    }
}

/* Shared by forward and backward: the EWA projection of Sigma3D (A.2 step 4). */
typedef struct {
    real tx, ty, tz;    /* clamped view-space mean */
    int x_in, y_in;     /* 1 when the un-clamped ratio was inside +-1.3 tanfov */
    real J00, J02, J11, J12;
    real T00, T01, T02, T10, T11, T12; /* T = J * Wr */
    real a, b, c;       /* Sigma2D + 0.3 I */
} ewa_t;

static void ewa_project(const real *pv, real fx, real fy, real tanfovx, real tanfovy, const real *V,
                        const real *S, ewa_t *e)
{
    real limx = R_(1.3) * tanfovx, limy = R_(1.3) * tanfovy;
    real txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e->x_in = !(txtz < -limx || txtz > limx);
    e->y_in = !(tytz < -limy || tytz > limy);
    e->tx = rmin(limx, rmax(-limx, txtz)) * pv[2];
    e->ty = rmin(limy, rmax(-limy, tytz)) * pv[2];
    e->tz = pv[2];
    e->J00 = fx / e->tz;
    e->J02 = -(fx * e->tx) / (e->tz * e->tz);
    e->J11 = fy / e->tz;
    e->J12 = -(fy * e->ty) / (e->tz * e->tz);
    /* Wr[i][j] = V[4*j + i] */
    e->T00 = e->J00 * V[0] + e->J02 * V[2];
    e->T01 = e->J00 * V[4] + e->J02 * V[6];
    e->T02 = e->J00 * V[8] + e->J02 * V[10];
    e->T10 = e->J11 * V[1] + e->J12 * V[2];
    e->T11 = e->J11 * V[5] + e->J12 * V[6];
    e->T12 = e->J11 * V[9] + e->J12 * V[10];
    /* u_r = Sigma * T_r^T */
    real u00 = S[0] * e->T00 + S[1] * e->T01 + S[2] * e->T02;
    real u01 = S[1] * e->T00 + S[3] * e->T01 + S[4] * e->T02;
    real u02 = S[2] * e->T00 + S[4] * e->T01 + S[5] * e->T02;
    real u10 = S[0] * e->T10 + S[1] * e->T11 + S[2] * e->T12;
    real u11 = S[1] * e->T10 + S[3] * e->T11 + S[4] * e->T12;
    real u12 = S[2] * e->T10 + S[4] * e->T11 + S[5] * e->T12;
    e->a = (e->T00 * u00 + e->T01 * u01 + e->T02 * u02) + R_(0.3);
    e->b = e->T00 * u10 + e->T01 * u11 + e->T02 * u12;
    e->c = (e->T10 * u10 + e->T11 * u11 + e->T12 * u12) + R_(0.3);
}

/* ------------------------------------------------------------------------------------ */
/* K1: per-Gaussian forward (A.2 + A.3).  All outputs are caller-allocated.             */
void oracle_preprocess(int P, int M, int D, int H, int W, real tanfovx, real tanfovy, real mod,
                       const real *means3D, const real *shs /*or NULL*/, const real *colors_precomp,
                       const real *opacities, const real *scales, const real *rots,
                       const real *cov3D_precomp /*or NULL*/, const real *V, const real *F,
                       const real *campos,
                       /* out */ real *depths, real *xy, real *conic_opacity, real *rgb, real *cov3D,
                       uint8_t *clamped, int32_t *radii, int32_t *rect, uint32_t *tiles_touched)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const real fx = (real)W / (R_(2) * tanfovx), fy = (real)H / (R_(2) * tanfovy);
    const int K = (D + 1) * (D + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {   /* (independent per Gaussian: every output element has one writer) */
        radii[i] = 0;
        tiles_touched[i] = 0;
        depths[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0, rect[4 * i + k] = 0;
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = 0, clamped[3 * i + k] = 0;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0;

        const real x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        real pv[3];
        pv[0] = V[0] * x + V[4] * y + V[8] * z + V[12];
        pv[1] = V[1] * x + V[5] * y + V[9] * z + V[13];
        pv[2] = V[2] * x + V[6] * y + V[10] * z + V[14];
        if (!(pv[2] > R_(0.2))) continue; /* near cull (NaN culls too) */

        real hx = F[0] * x + F[4] * y + F[8] * z + F[12];
        real hy = F[1] * x + F[5] * y + F[9] * z + F[13];
        real hw = F[3] * x + F[7] * y + F[11] * z + F[15];
        real pw = R_(1) / (hw + R_(0.0000001));
        real ndcx = hx * pw, ndcy = hy * pw;

        real S[6];
        if (cov3D_precomp) {
            for (int k = 0; k < 6; ++k) S[k] = cov3D_precomp[6 * i + k];
        } else {
            cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, S);
        }
        ewa_t e;
        ewa_project(pv, fx, fy, tanfovx, tanfovy, V, S, &e);
        real det = e.a * e.c - e.b * e.b;
        if (det == 0 || det != det) continue;
        /* The published algorithm culls det == 0 only: a projected covariance that is NOT positive
         * definite (det < 0, or a < 0 -- reachable only through a non-PSD cov3D_precomp; R S^2 R^T + 0.3 I never is) is
         * rasterized, and blends wherever its "power" happens to be <= 0 (a hyperbolic region of the image).  The HIP library
         * culls such a Gaussian (radius 0, include/hgs_rasterizer.h "Inputs the library rejects per Gaussian"): its blend
         * kernels evaluate the exponent through the conic's Cholesky factors, which exist only for a positive-definite
         * conic.  oracle_set_cull_non_pd(1) makes this checker apply the library's rule, so that radii / N / lists / image /
         * gradients can be compared exactly on such inputs; the default (0) is the published behaviour. */
        if (g_cull_non_pd && !(det > 0 && e.a > 0)) continue;
        real det_inv = R_(1) / det;
        real cx = e.c * det_inv, cy = -e.b * det_inv, cz = e.a * det_inv;
        real mid = R_(0.5) * (e.a + e.c);
        real sq = rsqrt_(rmax(R_(0.1), mid * mid - det));
        real l1 = mid + sq, l2 = mid - sq;
        real radf = rceil_(R_(3) * rsqrt_(rmax(l1, l2)));
        real px = ((ndcx + R_(1)) * (real)W - R_(1)) * R_(0.5);
        real py = ((ndcy + R_(1)) * (real)H - R_(1)) * R_(0.5);
        /* tile rect; clamp in the float domain, then truncate (identical to
           clamp(int(v),0,grid) for every finite v, and defined for inf/NaN) */
        real fminx = rmin((real)gx, rmax(R_(0), (px - radf) / R_(16)));
        real fmaxx = rmin((real)gx, rmax(R_(0), (px + radf + R_(15)) / R_(16)));
        real fminy = rmin((real)gy, rmax(R_(0), (py - radf) / R_(16)));
        real fmaxy = rmin((real)gy, rmax(R_(0), (py + radf + R_(15)) / R_(16)));
        /* non-finite centre or radius: culled (defined behaviour for inputs where upstream's
           float->int conversion would be undefined) */
        if (!(rabs_(px) <= R_(3.0e38)) || !(rabs_(py) <= R_(3.0e38)) || !(radf <= R_(1.0e9))) continue;
        int minx = (int)fminx, maxx = (int)fmaxx, miny = (int)fminy, maxy = (int)fmaxy;
        if (maxx <= minx || maxy <= miny) continue;

        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = S[k];
        if (shs) {
            real dx = x - campos[0], dy = y - campos[1], dz = z - campos[2];
            real len = rsqrt_(dx * dx + dy * dy + dz * dz);
            dx = dx / len, dy = dy / len, dz = dz / len;
            real B[16];
            sh_basis(D, dx, dy, dz, B);
            for (int c = 0; c < 3; ++c) {
                real acc = 0;
                for (int k = 0; k < K; ++k) acc += B[k] * shs[(size_t)i * M * 3 + 3 * k + c];
                acc += R_(0.5);
                clamped[3 * i + c] = (acc < 0);
                rgb[3 * i + c] = rmax(acc, R_(0));
            }
        } else {
            for (int c = 0; c < 3; ++c) rgb[3 * i + c] = colors_precomp[3 * i + c];
        }
        depths[i] = pv[2];
        radii[i] = (int32_t)radf;
        xy[2 * i] = px, xy[2 * i + 1] = py;
        conic_opacity[4 * i] = cx, conic_opacity[4 * i + 1] = cy, conic_opacity[4 * i + 2] = cz;
        conic_opacity[4 * i + 3] = opacities[i];
        rect[4 * i] = minx, rect[4 * i + 1] = miny, rect[4 * i + 2] = maxx, rect[4 * i + 3] = maxy;
        tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
    }
}

/* K2: inclusive scan; returns N */
int64_t oracle_scan(int P, const uint32_t *tiles_touched, uint32_t *offsets)
{
    uint64_t acc = 0;
    for (int i = 0; i < P; ++i) {
        acc += tiles_touched[i];
        offsets[i] = (uint32_t)acc;
    }
    return (int64_t)acc;
}

/* K3: key emission. depth bits are the raw fp32 pattern of the depth (always from the fp32
   value, also in the double build, so the key format is the same). */
void oracle_emit_keys(int P, int W, const real *depths, const int32_t *radii, const int32_t *rect,
                      const uint32_t *offsets, uint64_t *keys, uint32_t *values)
{
    const int gx = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {   /* (every Gaussian writes its own slots [offsets[i-1], offsets[i])) */
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : offsets[i - 1];
        float df = (float)depths[i];
        uint32_t dbits;
        memcpy(&dbits, &df, 4);
        for (int ty = rect[4 * i + 1]; ty < rect[4 * i + 3]; ++ty)
            for (int tx = rect[4 * i]; tx < rect[4 * i + 2]; ++tx) {
                uint64_t key = (uint64_t)(uint32_t)(ty * gx + tx);
                keys[off] = (key << 32) | dbits;
                values[off] = (uint32_t)i;
                ++off;
            }
    }
}

/* K4: stable LSD radix sort on the full 64-bit key (8 passes of 8 bits).  With OpenMP every thread takes one contiguous
   chunk of the array per pass: private digit counts, one prefix over (digit, thread) -- digit-major, so a digit's elements keep
   chunk order = input order --, then each thread scatters its own chunk in order: the same permutation as the serial loop. */
static int g_sort_chunks = 0;
void oracle_set_sort_chunks(int n) { g_sort_chunks = n; }
void oracle_sort_pairs(int64_t N, uint64_t *keys, uint32_t *values)
{
    if (N <= 1) return;
    uint64_t *k2 = (uint64_t *)malloc((size_t)N * 8);
    uint32_t *v2 = (uint32_t *)malloc((size_t)N * 4);
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = values, *vb = v2;
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
    if (nt > 64) nt = 64;            /* (256 counters per thread and pass: more chunks than this only lengthen the prefix) */
    if (N < ((int64_t)1 << 16)) nt = 1;
    if (g_sort_chunks > 0) nt = g_sort_chunks > 64 ? 64 : g_sort_chunks;   /* (tests: more chunks than the team has threads) */
#endif
    size_t *cnt = (size_t *)malloc(sizeof(size_t) * 256 * (size_t)nt);
    for (int pass = 0; pass < 8; ++pass) {
        const int sh = 8 * pass;
#pragma omp parallel num_threads(nt)
        {
            /* The array is cut into nt CHUNKS; the team may hold fewer threads than asked for (OMP_DYNAMIC, OMP_THREAD_LIMIT, a
               cgroup cap, a nested region): every thread takes chunks t, t + team, t + 2 team, ... so that none is left out. */
            int t = 0, team = 1;
#ifdef _OPENMP
            t = omp_get_thread_num(), team = omp_get_num_threads();
#endif
            for (int ch = t; ch < nt; ch += team) {
                const int64_t lo = N * ch / nt, hi = N * (ch + 1) / nt;
                size_t *c = cnt + 256 * (size_t)ch;
                memset(c, 0, sizeof(size_t) * 256);
                for (int64_t i = lo; i < hi; ++i) c[(ka[i] >> sh) & 255]++;
            }
#pragma omp barrier
#pragma omp single
            {
                size_t pos = 0;
                for (int d = 0; d < 256; ++d)
                    for (int u = 0; u < nt; ++u) {
                        const size_t n = cnt[256 * (size_t)u + d];
                        cnt[256 * (size_t)u + d] = pos;
                        pos += n;
                    }
            } /* (implicit barrier) */
            for (int ch = t; ch < nt; ch += team) {
                const int64_t lo = N * ch / nt, hi = N * (ch + 1) / nt;
                size_t *c = cnt + 256 * (size_t)ch;
                for (int64_t i = lo; i < hi; ++i) {
                    const size_t p = c[(ka[i] >> sh) & 255]++;
                    kb[p] = ka[i], vb[p] = va[i];
                }
            }
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    /* 8 passes: data is back in the caller's arrays */
    free(cnt);
    free(k2);
    free(v2);
}

/* K5: tile ranges [start,end) */
void oracle_tile_ranges(int64_t N, const uint64_t *keys, int num_tiles, uint32_t *ranges)
{
    memset(ranges, 0, (size_t)num_tiles * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {   /* (a tile's start / end is written by the one element that sees the boundary) */
        uint32_t t = (uint32_t)(keys[i] >> 32);
        if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) ranges[2 * t] = (uint32_t)i;
        if (i == N - 1 || (uint32_t)(keys[i + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
}

/* K6: front-to-back blend (A.4) */
void oracle_blend_forward(int H, int W, const uint32_t *ranges, const uint32_t *values, const real *xy,
                          const real *conic_opacity, const real *rgb, const real *bg,
                          real *out_color, real *final_T, uint32_t *n_contrib)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                real T = 1, C0 = 0, C1 = 0, C2 = 0;
                uint32_t contributor = 0, last = 0;
                for (uint32_t j = s; j < e; ++j) {
                    ++contributor;
                    const uint32_t g = values[j];
                    const real dx = xy[2 * g] - (real)px, dy = xy[2 * g + 1] - (real)py;
                    const real *co = conic_opacity + 4 * g;
                    const real power = R_(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0) continue;
                    const real alpha = rmin(R_(0.99), co[3] * rexp_(power));
                    if (alpha < R_(1.0) / R_(255.0)) continue;
                    const real test_T = T * (R_(1) - alpha);
                    if (test_T < R_(0.0001)) break;
                    const real w = alpha * T;
                    C0 += rgb[3 * g] * w, C1 += rgb[3 * g + 1] * w, C2 += rgb[3 * g + 2] * w;
                    T = test_T;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                final_T[pix] = T;
                n_contrib[pix] = last;
                out_color[0 * (size_t)H * W + pix] = C0 + T * bg[0];
                out_color[1 * (size_t)H * W + pix] = C1 + T * bg[1];
                out_color[2 * (size_t)H * W + pix] = C2 + T * bg[2];
            }
    }
}

/* K7: back-to-front pixel backward (A.5 pixel part).  Sums are accumulated in double so the checker is as
   order-insensitive as possible; outputs are rounded to `real` once.  A tile sums its 256 pixels' contributions per LIST
   ENTRY in a tile-local array first and then adds each entry's nine sums to the Gaussian's shared double accumulator with
   `omp atomic` -- one P x 9 array whatever the thread count (round 3 kept one per thread: 461 MB at 32 threads, which is
   what capped the CPU baseline at 32 of the host's cores).  Outputs are OVERWRITTEN.
   dL_dmean2D [P,3] (z = 0), dL_dconic [P,4] (slots x,y,.,w), dL_dopacity [P], dL_dcolor [P,3] */
void oracle_blend_backward(int P, int H, int W, const uint32_t *ranges, const uint32_t *values,
                           const real *xy, const real *conic_opacity, const real *rgb, const real *bg,
                           const real *final_T, const uint32_t *n_contrib, const real *dL_dpix,
                           real *dL_dmean2D, real *dL_dconic, real *dL_dopacity, real *dL_dcolor)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    double *acc = (double *)calloc((size_t)P * 9 + 1, sizeof(double));
    const real ddelx_dx = R_(0.5) * (real)W, ddely_dy = R_(0.5) * (real)H;
#pragma omp parallel
    {
        double *loc = NULL;   /* this thread's tile-local sums: [entries of the tile][9], grown on demand */
        size_t loc_cap = 0;
#pragma omp for schedule(dynamic, 4)
        for (int tile = 0; tile < gx * gy; ++tile) {
            const int tx = tile % gx, ty = tile / gx;
            const uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
            if (e <= s) continue;
            const size_t n_tile = (size_t)(e - s);
            if (n_tile > loc_cap) {
                free(loc);
                loc_cap = n_tile + n_tile / 2;
                loc = (double *)malloc(loc_cap * 9 * sizeof(double));
            }
            memset(loc, 0, n_tile * 9 * sizeof(double));
            uint32_t deepest = 0;   /* entries beyond the deepest n_contrib of the tile received nothing */
            for (int ly = 0; ly < TILE; ++ly)
                for (int lx = 0; lx < TILE; ++lx) {
                    const int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    const size_t pix = (size_t)py * W + px;
                    const real T_final = final_T[pix];
                    real T = T_final;
                    const real g0 = dL_dpix[pix], g1 = dL_dpix[(size_t)H * W + pix], g2 = dL_dpix[2 * (size_t)H * W + pix];
                    const real bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
                    real ar0 = 0, ar1 = 0, ar2 = 0, lc0 = 0, lc1 = 0, lc2 = 0, last_alpha = 0;
                    if (n_contrib[pix] > deepest) deepest = n_contrib[pix];
                    for (int64_t j = (int64_t)s + (int64_t)n_contrib[pix] - 1; j >= (int64_t)s; --j) {
                        const uint32_t g = values[j];
                        const real dx = xy[2 * g] - (real)px, dy = xy[2 * g + 1] - (real)py;
                        const real *co = conic_opacity + 4 * g;
                        const real power = R_(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0) continue;
                        const real G = rexp_(power);
                        const real alpha = rmin(R_(0.99), co[3] * G);
                        if (alpha < R_(1.0) / R_(255.0)) continue;
                        T = T / (R_(1) - alpha);
                        const real dch = alpha * T;
                        const real c0 = rgb[3 * g], c1 = rgb[3 * g + 1], c2 = rgb[3 * g + 2];
                        ar0 = last_alpha * lc0 + (R_(1) - last_alpha) * ar0;
                        ar1 = last_alpha * lc1 + (R_(1) - last_alpha) * ar1;
                        ar2 = last_alpha * lc2 + (R_(1) - last_alpha) * ar2;
                        lc0 = c0, lc1 = c1, lc2 = c2;
                        real dL_dalpha = (c0 - ar0) * g0 + (c1 - ar1) * g1 + (c2 - ar2) * g2;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (R_(1) - alpha)) * bg_dot;
                        const real dL_dG = co[3] * dL_dalpha;
                        const real gdx = G * dx, gdy = G * dy;
                        const real dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const real dG_ddely = -gdy * co[2] - gdx * co[1];
                        double *a = loc + (size_t)(j - (int64_t)s) * 9;
                        a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        a[2] += (double)(R_(-0.5) * gdx * dx * dL_dG);
                        a[3] += (double)(R_(-0.5) * gdx * dy * dL_dG);
                        a[4] += (double)(R_(-0.5) * gdy * dy * dL_dG);
                        a[5] += (double)(G * dL_dalpha);
                        a[6] += (double)(dch * g0);
                        a[7] += (double)(dch * g1);
                        a[8] += (double)(dch * g2);
                    }
                }
            for (uint32_t k = 0; k < deepest; ++k) {
                const double *a = loc + (size_t)k * 9;
                double *dst = acc + (size_t)values[s + k] * 9;
                for (int c = 0; c < 9; ++c)
                    if (a[c] != 0.0) {
#pragma omp atomic
                        dst[c] += a[c];
                    }
            }
        }
        free(loc);
    }
    for (int i = 0; i < P; ++i) {
        const double *t = acc + (size_t)i * 9;
        dL_dmean2D[3 * i] = (real)t[0], dL_dmean2D[3 * i + 1] = (real)t[1], dL_dmean2D[3 * i + 2] = 0;
        dL_dconic[4 * i] = (real)t[2], dL_dconic[4 * i + 1] = (real)t[3], dL_dconic[4 * i + 2] = 0;
        dL_dconic[4 * i + 3] = (real)t[4];
        dL_dopacity[i] = (real)t[5];
        dL_dcolor[3 * i] = (real)t[6], dL_dcolor[3 * i + 1] = (real)t[7], dL_dcolor[3 * i + 2] = (real)t[8];
    }
    free(acc);
}

/* K8+K9: per-Gaussian backward (A.5 Gaussian part).  Outputs OVERWRITTEN (zeros where
   radius == 0).  dL_dcov3D [P,6] is returned too (it is the gradient of cov3D_precomp
   when that input is used). */
void oracle_preprocess_backward(int P, int M, int D, int H, int W, real tanfovx, real tanfovy, real mod,
                                const real *means3D, const real *shs, const real *scales, const real *rots,
                                const real *cov3D_precomp, const real *V, const real *F, const real *campos,
                                const int32_t *radii, const real *cov3D, const uint8_t *clamped,
                                const real *dL_dmean2D, const real *dL_dconic, const real *dL_dcolor,
                                /* out */ real *dL_dmeans3D, real *dL_dsh, real *dL_dscale, real *dL_drot,
                                real *dL_dcov3D)
{
    const real fx = (real)W / (R_(2) * tanfovx), fy = (real)H / (R_(2) * tanfovy);
    const int K = (D + 1) * (D + 1);
    memset(dL_dmeans3D, 0, sizeof(real) * 3 * (size_t)P);
    if (dL_dsh) memset(dL_dsh, 0, sizeof(real) * 3 * (size_t)M * (size_t)P);
    memset(dL_dscale, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_drot, 0, sizeof(real) * 4 * (size_t)P);
    memset(dL_dcov3D, 0, sizeof(real) * 6 * (size_t)P);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {   /* (independent per Gaussian) */
        if (radii[i] <= 0) continue;
        const real x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        real dmean[3] = {0, 0, 0};

        /* ---- conic -> Sigma2D -> Sigma3D, and -> view-space mean via J ---- */
        real pv[3];
        pv[0] = V[0] * x + V[4] * y + V[8] * z + V[12];
        pv[1] = V[1] * x + V[5] * y + V[9] * z + V[13];
        pv[2] = V[2] * x + V[6] * y + V[10] * z + V[14];
        const real *S = cov3D + 6 * i;
        ewa_t e;
        ewa_project(pv, fx, fy, tanfovx, tanfovy, V, S, &e);
        const real a = e.a, b = e.b, c = e.c;
        const real gxx = dL_dconic[4 * i], gxy = dL_dconic[4 * i + 1], gyy = dL_dconic[4 * i + 3];
        const real denom = a * c - b * b;
        const real d2inv = R_(1) / (denom * denom + R_(0.0000001));
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real *dS = dL_dcov3D + 6 * i;
        if (d2inv != 0) {
            dL_da = d2inv * (-c * c * gxx + R_(2) * b * c * gxy + (denom - a * c) * gyy);
            dL_dc = d2inv * (-a * a * gyy + R_(2) * a * b * gxy + (denom - a * c) * gxx);
            dL_db = d2inv * R_(2) * (b * c * gxx - (denom + R_(2) * b * b) * gxy + a * b * gyy);
            dS[0] = e.T00 * e.T00 * dL_da + e.T00 * e.T10 * dL_db + e.T10 * e.T10 * dL_dc;
            dS[3] = e.T01 * e.T01 * dL_da + e.T01 * e.T11 * dL_db + e.T11 * e.T11 * dL_dc;
            dS[5] = e.T02 * e.T02 * dL_da + e.T02 * e.T12 * dL_db + e.T12 * e.T12 * dL_dc;
            dS[1] = R_(2) * e.T00 * e.T01 * dL_da + (e.T00 * e.T11 + e.T01 * e.T10) * dL_db + R_(2) * e.T10 * e.T11 * dL_dc;
            dS[2] = R_(2) * e.T00 * e.T02 * dL_da + (e.T00 * e.T12 + e.T02 * e.T10) * dL_db + R_(2) * e.T10 * e.T12 * dL_dc;
            dS[4] = R_(2) * e.T02 * e.T01 * dL_da + (e.T01 * e.T12 + e.T02 * e.T11) * dL_db + R_(2) * e.T11 * e.T12 * dL_dc;
        }
        /* dL/dT (2x3): Sigma2D = T Sigma T^T */
        real u00 = S[0] * e.T00 + S[1] * e.T01 + S[2] * e.T02;
        real u01 = S[1] * e.T00 + S[3] * e.T01 + S[4] * e.T02;
        real u02 = S[2] * e.T00 + S[4] * e.T01 + S[5] * e.T02;
        real u10 = S[0] * e.T10 + S[1] * e.T11 + S[2] * e.T12;
        real u11 = S[1] * e.T10 + S[3] * e.T11 + S[4] * e.T12;
        real u12 = S[2] * e.T10 + S[4] * e.T11 + S[5] * e.T12;
        real dT00 = R_(2) * u00 * dL_da + u10 * dL_db, dT01 = R_(2) * u01 * dL_da + u11 * dL_db,
             dT02 = R_(2) * u02 * dL_da + u12 * dL_db;
        real dT10 = R_(2) * u10 * dL_dc + u00 * dL_db, dT11 = R_(2) * u11 * dL_dc + u01 * dL_db,
             dT12 = R_(2) * u12 * dL_dc + u02 * dL_db;
        real dJ00 = V[0] * dT00 + V[4] * dT01 + V[8] * dT02;
        real dJ02 = V[2] * dT00 + V[6] * dT01 + V[10] * dT02;
        real dJ11 = V[1] * dT10 + V[5] * dT11 + V[9] * dT12;
        real dJ12 = V[2] * dT10 + V[6] * dT11 + V[10] * dT12;
        real tz = R_(1) / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        real dtx = (e.x_in ? R_(1) : R_(0)) * (-fx * tz2 * dJ02);
        real dty = (e.y_in ? R_(1) : R_(0)) * (-fy * tz2 * dJ12);
        real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (R_(2) * fx * e.tx) * tz3 * dJ02 + (R_(2) * fy * e.ty) * tz3 * dJ12;
        /* dmean = Wr^T dt ; Wr[i][k] = V[4k+i] */
        dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;

        /* ---- mean2D (NDC-scaled) -> mean through the perspective divide ---- */
        {
            real hx = F[0] * x + F[4] * y + F[8] * z + F[12];
            real hy = F[1] * x + F[5] * y + F[9] * z + F[13];
            real hw = F[3] * x + F[7] * y + F[11] * z + F[15];
            real mw = R_(1) / (hw + R_(0.0000001));
            real mul1 = hx * mw * mw, mul2 = hy * mw * mw;
            real g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
            dmean[0] += (F[0] * mw - F[3] * mul1) * g2x + (F[1] * mw - F[3] * mul2) * g2y;
            dmean[1] += (F[4] * mw - F[7] * mul1) * g2x + (F[5] * mw - F[7] * mul2) * g2y;
            dmean[2] += (F[8] * mw - F[11] * mul1) * g2x + (F[9] * mw - F[11] * mul2) * g2y;
        }

        /* ---- SH backward ---- */
        if (shs) {
            real dr[3];
            for (int ch = 0; ch < 3; ++ch) dr[ch] = clamped[3 * i + ch] ? R_(0) : dL_dcolor[3 * i + ch];
            real vx = x - campos[0], vy = y - campos[1], vz = z - campos[2];
            real len = rsqrt_(vx * vx + vy * vy + vz * vz);
            real dxn = vx / len, dyn = vy / len, dzn = vz / len;
            real B[16];
            sh_basis(D, dxn, dyn, dzn, B);
            const real *sh = shs + (size_t)i * M * 3;
            real *dsh = dL_dsh + (size_t)i * M * 3;
            for (int k = 0; k < K; ++k)
                for (int ch = 0; ch < 3; ++ch) dsh[3 * k + ch] = B[k] * dr[ch];
            /* dRGB/d(dir) . dr, per basis derivative */
            real dBx[16] = {0}, dBy[16] = {0}, dBz[16] = {0};
            if (D > 0) {
                dBy[1] = -R_(SH_C1); dBz[2] = R_(SH_C1); dBx[3] = -R_(SH_C1);
                if (D > 1) {
                    real X = dxn, Y = dyn, Z = dzn, xx = X * X, yy = Y * Y, zz = Z * Z;
                    dBx[4] = R_(SH_C2[0]) * Y; dBy[4] = R_(SH_C2[0]) * X;
                    dBy[5] = R_(SH_C2[1]) * Z; dBz[5] = R_(SH_C2[1]) * Y;
                    dBx[6] = R_(SH_C2[2]) * R_(-2) * X; dBy[6] = R_(SH_C2[2]) * R_(-2) * Y; dBz[6] = R_(SH_C2[2]) * R_(4) * Z;
                    dBx[7] = R_(SH_C2[3]) * Z; dBz[7] = R_(SH_C2[3]) * X;
                    dBx[8] = R_(SH_C2[4]) * R_(2) * X; dBy[8] = R_(SH_C2[4]) * R_(-2) * Y;
                    if (D > 2) {
                        dBx[9] = R_(SH_C3[0]) * R_(6) * X * Y; dBy[9] = R_(SH_C3[0]) * (R_(3) * xx - R_(3) * yy);
                        dBx[10] = R_(SH_C3[1]) * Y * Z; dBy[10] = R_(SH_C3[1]) * X * Z; dBz[10] = R_(SH_C3[1]) * X * Y;
                        dBx[11] = R_(SH_C3[2]) * R_(-2) * X * Y; dBy[11] = R_(SH_C3[2]) * (R_(4) * zz - xx - R_(3) * yy);
                        dBz[11] = R_(SH_C3[2]) * R_(8) * Y * Z;
                        dBx[12] = R_(SH_C3[3]) * R_(-6) * X * Z; dBy[12] = R_(SH_C3[3]) * R_(-6) * Y * Z;
                        dBz[12] = R_(SH_C3[3]) * (R_(6) * zz - R_(3) * xx - R_(3) * yy);
                        dBx[13] = R_(SH_C3[4]) * (R_(4) * zz - R_(3) * xx - yy); dBy[13] = R_(SH_C3[4]) * R_(-2) * X * Y;
                        dBz[13] = R_(SH_C3[4]) * R_(8) * X * Z;
                        dBx[14] = R_(SH_C3[5]) * R_(2) * X * Z; dBy[14] = R_(SH_C3[5]) * R_(-2) * Y * Z;
                        dBz[14] = R_(SH_C3[5]) * (xx - yy);
                        dBx[15] = R_(SH_C3[6]) * (R_(3) * xx - R_(3) * yy); dBy[15] = R_(SH_C3[6]) * R_(-6) * X * Y;
                    }
                }
            }
            real ddx = 0, ddy = 0, ddz = 0;
            for (int k = 1; k < K; ++k) {
                real w = sh[3 * k] * dr[0] + sh[3 * k + 1] * dr[1] + sh[3 * k + 2] * dr[2];
                ddx += dBx[k] * w, ddy += dBy[k] * w, ddz += dBz[k] * w;
            }
            /* back through d = v/|v| */
            real s2 = vx * vx + vy * vy + vz * vz;
            real inv32 = R_(1) / rsqrt_(s2 * s2 * s2);
            dmean[0] += ((s2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * inv32;
            dmean[1] += (-vx * vy * ddx + (s2 - vy * vy) * ddy - vz * vy * ddz) * inv32;
            dmean[2] += (-vx * vz * ddx - vy * vz * ddy + (s2 - vz * vz) * ddz) * inv32;
        }
        dL_dmeans3D[3 * i] = dmean[0], dL_dmeans3D[3 * i + 1] = dmean[1], dL_dmeans3D[3 * i + 2] = dmean[2];

        /* ---- Sigma3D -> scale, quaternion ---- */
        if (!cov3D_precomp) {
            const real *q = rots + 4 * i;
            real r = q[0], qx = q[1], qy = q[2], qz = q[3];
            real s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
            real R[3][3] = {{R_(1) - R_(2) * (qy * qy + qz * qz), R_(2) * (qx * qy - r * qz), R_(2) * (qx * qz + r * qy)},
                            {R_(2) * (qx * qy + r * qz), R_(1) - R_(2) * (qx * qx + qz * qz), R_(2) * (qy * qz - r * qx)},
                            {R_(2) * (qx * qz - r * qy), R_(2) * (qy * qz + r * qx), R_(1) - R_(2) * (qx * qx + qy * qy)}};
            real Gs[3][3] = {{dS[0], R_(0.5) * dS[1], R_(0.5) * dS[2]},
                             {R_(0.5) * dS[1], dS[3], R_(0.5) * dS[4]},
                             {R_(0.5) * dS[2], R_(0.5) * dS[4], dS[5]}};
            real dM[3][3], dR[3][3];
            for (int ii = 0; ii < 3; ++ii)
                for (int jj = 0; jj < 3; ++jj) {
                    real accm = 0;
                    for (int kk = 0; kk < 3; ++kk) accm += Gs[ii][kk] * (R[kk][jj] * s[jj]);
                    dM[ii][jj] = R_(2) * accm;
                }
            for (int jj = 0; jj < 3; ++jj) {
                real ds = R[0][jj] * dM[0][jj] + R[1][jj] * dM[1][jj] + R[2][jj] * dM[2][jj];
                /* the true derivative carries the scale modifier; the published kernel omits the factor (SURVEY.md A.5 note):
                 * oracle_set_upstream_scale_grad(1) reproduces that convention */
                dL_dscale[3 * i + jj] = (g_upstream_scale_grad ? (real)1 : mod) * ds;
                for (int ii = 0; ii < 3; ++ii) dR[ii][jj] = s[jj] * dM[ii][jj];
            }
            dL_drot[4 * i + 0] = R_(2) * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
            dL_drot[4 * i + 1] = R_(2) * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - R_(2) * qx * dR[1][1] - r * dR[1][2] + qz * dR[2][0] + r * dR[2][1] - R_(2) * qx * dR[2][2]);
            dL_drot[4 * i + 2] = R_(2) * (-R_(2) * qy * dR[0][0] + qx * dR[0][1] + r * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] - r * dR[2][0] + qz * dR[2][1] - R_(2) * qy * dR[2][2]);
            dL_drot[4 * i + 3] = R_(2) * (-R_(2) * qz * dR[0][0] - r * dR[0][1] + qx * dR[0][2] + r * dR[1][0] - R_(2) * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        }
    }
}

/* K10: markVisible */
void oracle_mark_visible(int P, const real *means3D, const real *V, uint8_t *present)
{
    for (int i = 0; i < P; ++i) {
        real z = V[2] * means3D[3 * i] + V[6] * means3D[3 * i + 1] + V[10] * means3D[3 * i + 2] + V[14];
        present[i] = z > R_(0.2);
    }
}

void oracle_set_upstream_scale_grad(int on) { g_upstream_scale_grad = on; }
void oracle_set_cull_non_pd(int on) { g_cull_non_pd = on; }

void oracle_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
