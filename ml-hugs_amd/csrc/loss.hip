// Row f-5 (the consumer on the other side of the rasterizer's output): the photometric loss of every training step,
//   l1_loss(pred, gt)            /root/reference/hugs/losses/utils.py:54-58
//   ssim(pred, gt)               /root/reference/hugs/losses/utils.py:65-108 (11x11 Gaussian window, sigma 1.5, zero padding,
//                                groups = channels, C1 = 0.01^2, C2 = 0.03^2, mean over everything)
// called on the full render and again on the human-only render (hugs/losses/loss.py:88-107,128-137).  The reference spends
// five depthwise 11x11 conv2d + ~15 elementwise kernels on it forward and the same again backward, on 1080p images: more GPU
// time than the rasterizer's own forward + backward.  Here: one forward kernel (both images read once, the five windowed
// moments through LDS, separable: 11 + 11 taps instead of 121), one 256-thread reduction, one backward kernel.
//
// Forward, per pixel (mu1 = w*x, mu2 = w*y, E11 = w*x^2, E22 = w*y^2, E12 = w*xy; * = zero-padded correlation):
//   A = 2 mu1 mu2 + C1,  B = 2 (E12 - mu1 mu2) + C2,  Cc = mu1^2 + mu2^2 + C1,  D = (E11 - mu1^2) + (E22 - mu2^2) + C2
//   map = A B / (Cc D)
// and, for the backward (x = pred is the only differentiable input, as at every reference call site), the three partials
//   m1 = dmap/dmu1 = 2 mu2 (B - A)/(Cc D) - 2 mu1 map (1/Cc - 1/D),   m2 = dmap/dE11 = -map / D,   m3 = dmap/dE12 = 2 A/(Cc D)
// are stored; backward is the adjoint of the three correlations (the window is symmetric):
//   dL/dx = g_ssim/(C H W) [ w*m1 + 2 x (w*m2) + y (w*m3) ] + g_l1 sign(x - y)
// Tiles of 64 x 16 pixels per 256-thread workgroup, halo 5.  Both kernels move ~4 bytes per pixel and quantity once; the
// arithmetic (110 FMAs per pixel forward) and the LDS traffic are what the time goes into, not HBM.
// Sums: per-workgroup partials, added up in a fixed order in double by the reduction kernel (no float atomics: the same loss
// bit for bit on every run).
#include "hgs_common.h"

namespace {

constexpr int SSIM_R = 5, SSIM_TW = 64, SSIM_TH = 16, SSIM_IW = SSIM_TW + 2 * SSIM_R, SSIM_IH = SSIM_TH + 2 * SSIM_R;
constexpr int SSIM_SEG = 8;  // columns per thread in the horizontal pass
// gauss(11, 1.5) / sum, in fp32 as the reference builds it (utils.py:65-67)
__host__ __device__ constexpr float ssim_w(int k)  // (a function, so that the unrolled loops see literals)
{
    constexpr float W[6] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f, 2.130055279e-01f, 2.660117149e-01f};
    return W[k < 6 ? k : 10 - k];
}

struct Plane {
    int C, H, W;
    __device__ __forceinline__ size_t at(int c, int y, int x) const { return ((size_t)c * H + y) * W + x; }
};

// LDS tile: rows of SSIM_LW floats, image column x0 - SSIM_PAD + c at position c -- the halo (5) is padded to 8 so that a row
// starts on a 16-byte boundary of the image row and is fetched as float4s (when W is a multiple of 4 and the plane is 16-byte
// aligned; x0 is a multiple of 64)
constexpr int SSIM_PAD = 8, SSIM_LW = SSIM_TW + 2 * SSIM_PAD, SSIM_OFF = SSIM_PAD - SSIM_R;  // 80 columns; the window starts at column 3
typedef float TileRow[SSIM_LW + 1];

// Workgroup -> tile.  Consecutive workgroups are dealt round-robin to the 8 XCDs, each with its own L2; a tile shares 5-pixel halos
// with its neighbours, so every XCD gets a contiguous BAND of the (channel, row, column) tile order instead of every 8th tile:
// workgroup b works on tile (b % 8) * ceil(T / 8) + b / 8 (the grid is 8 * ceil(T / 8) workgroups; the surplus leaves at once).
// 1080p: forward 57.7 -> 53.5 us, backward 49.1 -> 37.4 us.
struct TileId { int ch, x0, y0, linear; bool valid; };
__device__ __forceinline__ TileId tile_of_workgroup(int tiles_x, int tiles_y, int C)
{
    const int T = tiles_x * tiles_y * C, per = (T + 7) / 8;
    const int t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    const int ch = t / (tiles_x * tiles_y), r = t - ch * (tiles_x * tiles_y), ty = r / tiles_x;
    return {ch, (r - ty * tiles_x) * SSIM_TW, ty * SSIM_TH, t, t < T && (int)(blockIdx.x >> 3) < per};
}

// loads the tile + halo of one channel (`src`: the channel's H x W plane) into LDS, zero outside the image
__device__ __forceinline__ void load_tile(TileRow* dst, int x0, int y0, int H, int W, const float* __restrict__ src)
{
    if ((W & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        for (int idx = threadIdx.x; idx < SSIM_IH * (SSIM_LW / 4); idx += 256) {
            const int r = idx / (SSIM_LW / 4), c = (idx - r * (SSIM_LW / 4)) * 4;
            const int gy = y0 - SSIM_R + r, gx = x0 - SSIM_PAD + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(src + (size_t)gy * W + gx);
            dst[r][c] = v.x, dst[r][c + 1] = v.y, dst[r][c + 2] = v.z, dst[r][c + 3] = v.w;
        }
    } else {
        for (int idx = threadIdx.x; idx < SSIM_IH * SSIM_IW; idx += 256) {
            const int r = idx / SSIM_IW, c = idx - r * SSIM_IW;
            const int gy = y0 - SSIM_R + r, gx = x0 - SSIM_R + c;
            dst[r][c + SSIM_OFF] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.0f;
        }
    }
}

template <bool WITH_MAPS>
__global__ void __launch_bounds__(256)
ssim_l1_forward_kernel(Plane p, const float* __restrict__ img1, const float* __restrict__ img2, float* __restrict__ maps,
                       float2* __restrict__ partial)
{
    // the two image tiles stay in LDS; the five windowed moments go one at a time through ONE row-sum buffer (24 KB of LDS
    // instead of 51, registers for one moment at a time: six workgroups per CU instead of three)
    __shared__ TileRow sx[SSIM_IH], sy[SSIM_IH];
    __shared__ float hq[SSIM_IH][SSIM_TW + 1];
    __shared__ float2 wsum[4];
    const TileId tile = tile_of_workgroup((p.W + SSIM_TW - 1) / SSIM_TW, (p.H + SSIM_TH - 1) / SSIM_TH, p.C);
    if (!tile.valid) return;  // (uniform)
    const int ch = tile.ch, x0 = tile.x0, y0 = tile.y0, tid = threadIdx.x;
    load_tile(sx, x0, y0, p.H, p.W, img1 + (size_t)ch * p.H * p.W);
    load_tile(sy, x0, y0, p.H, p.W, img2 + (size_t)ch * p.H * p.W);
    const int col = tid & (SSIM_TW - 1), r0 = (tid / SSIM_TW) * 4;
    float acc[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        __syncthreads();  // (the tiles are loaded / the previous moment's vertical pass is done with hq)
        // horizontal pass: a thread = one row, 8 adjacent columns (18 inputs in registers)
        if (tid < SSIM_IH * (SSIM_TW / SSIM_SEG)) {
            const int row = tid / (SSIM_TW / SSIM_SEG), c0 = (tid - row * (SSIM_TW / SSIM_SEG)) * SSIM_SEG;
            float v[SSIM_SEG + 2 * SSIM_R];
#pragma unroll
            for (int j = 0; j < SSIM_SEG + 2 * SSIM_R; ++j) {
                const float xv = sx[row][c0 + j + SSIM_OFF], yv = sy[row][c0 + j + SSIM_OFF];
                v[j] = q == 0 ? xv : q == 1 ? yv : q == 2 ? xv * xv : q == 3 ? yv * yv : xv * yv;
            }
#pragma unroll
            for (int o = 0; o < SSIM_SEG; ++o) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) t = __builtin_fmaf(ssim_w(k), v[o + k], t);
                hq[row][c0 + o] = t;
            }
        }
        __syncthreads();
        // vertical pass: a thread = one column, 4 adjacent rows
        float v[4 + 2 * SSIM_R];
#pragma unroll
        for (int j = 0; j < 4 + 2 * SSIM_R; ++j) v[j] = hq[r0 + j][col];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) t = __builtin_fmaf(ssim_w(k), v[o + k], t);
            acc[q][o] = t;
        }
    }
    const float C1 = 0.0001f, C2 = 0.0009f;
    float ssim_sum = 0.f, l1_sum = 0.f;
    const int gx = x0 + col;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + r0 + o;
        if (gx < p.W && gy < p.H) {
            const float mu1 = acc[0][o], mu2 = acc[1][o], mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = acc[2][o] - mu1_sq, s2 = acc[3][o] - mu2_sq, s12 = acc[4][o] - mu12;
            const float A = 2.0f * mu12 + C1, B = 2.0f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
            const float map = (A * B) / (Cc * D);
            ssim_sum += map;
            l1_sum += fabsf(sx[r0 + o + SSIM_R][col + SSIM_PAD] - sy[r0 + o + SSIM_R][col + SSIM_PAD]);
            if (WITH_MAPS) {
                // (the map itself is the reference's expression, correctly rounded; its partials take the 1-ulp reciprocals)
                const float inv_c = __builtin_amdgcn_rcpf(Cc), inv_d = __builtin_amdgcn_rcpf(D), inv_cd = inv_c * inv_d;
                const size_t at = p.at(ch, gy, gx), plane = (size_t)p.C * p.H * p.W;
                maps[at] = 2.0f * mu2 * (B - A) * inv_cd - 2.0f * mu1 * map * (inv_c - inv_d);
                maps[plane + at] = -map * inv_d;
                maps[2 * plane + at] = 2.0f * A * inv_cd;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ssim_sum += __shfl_xor(ssim_sum, d, 64), l1_sum += __shfl_xor(l1_sum, d, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = make_float2(ssim_sum, l1_sum);
    __syncthreads();
    if (tid == 0) {
        const float2 s = make_float2((wsum[0].x + wsum[1].x) + (wsum[2].x + wsum[3].x), (wsum[0].y + wsum[1].y) + (wsum[2].y + wsum[3].y));
        partial[tile.linear] = s;
    }
}

// out[0] = mean of the SSIM map, out[1] = mean |x - y|, out[2] = sum |x - y| (l1_loss with a mask divides it by mask.sum())
__global__ void __launch_bounds__(256) ssim_l1_reduce_kernel(int blocks, const float2* __restrict__ partial, double count, float* __restrict__ out)
{
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) a += (double)partial[i].x, b += (double)partial[i].y;
    sa[threadIdx.x] = a, sb[threadIdx.x] = b;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sa[threadIdx.x] += sa[threadIdx.x + d], sb[threadIdx.x] += sb[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sa[0] / count), out[1] = (float)(sb[0] / count), out[2] = (float)sb[0];
}

__global__ void __launch_bounds__(256)
ssim_l1_backward_kernel(Plane p, const float* __restrict__ img1, const float* __restrict__ img2, const float* __restrict__ maps,
                        const float* __restrict__ g_ssim_mean, const float* __restrict__ g_l1_sum, float* __restrict__ dL_dimg1)
{
    // one quantity at a time through ONE tile and ONE row-sum buffer (15 KB of LDS instead of 46: eight workgroups per CU instead
    // of three -- this kernel waits on memory, not on arithmetic)
    __shared__ TileRow sm[SSIM_IH];
    __shared__ float hq[SSIM_IH][SSIM_TW + 1];
    const TileId tile = tile_of_workgroup((p.W + SSIM_TW - 1) / SSIM_TW, (p.H + SSIM_TH - 1) / SSIM_TH, p.C);
    if (!tile.valid) return;  // (uniform)
    const int ch = tile.ch, x0 = tile.x0, y0 = tile.y0, tid = threadIdx.x;
    const size_t plane = (size_t)p.C * p.H * p.W;
    const float gs = g_ssim_mean ? g_ssim_mean[0] / (float)((double)p.C * p.H * p.W) : 0.0f, gl = g_l1_sum ? g_l1_sum[0] : 0.0f;
    const int col = tid & (SSIM_TW - 1), r0 = (tid / SSIM_TW) * 4;
    float acc[3][4] = {};
    if (maps) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            load_tile(sm, x0, y0, p.H, p.W, maps + q * plane + (size_t)ch * p.H * p.W);
            __syncthreads();  // (also: the previous quantity's vertical pass is done with hq)
            if (tid < SSIM_IH * (SSIM_TW / SSIM_SEG)) {
                const int row = tid / (SSIM_TW / SSIM_SEG), c0 = (tid - row * (SSIM_TW / SSIM_SEG)) * SSIM_SEG;
                float v[SSIM_SEG + 2 * SSIM_R];
#pragma unroll
                for (int j = 0; j < SSIM_SEG + 2 * SSIM_R; ++j) v[j] = sm[row][c0 + j + SSIM_OFF];
#pragma unroll
                for (int o = 0; o < SSIM_SEG; ++o) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) t = __builtin_fmaf(ssim_w(k), v[o + k], t);
                    hq[row][c0 + o] = t;
                }
            }
            __syncthreads();  // (also: the horizontal pass is done with sm, the next quantity may overwrite it)
            float v[4 + 2 * SSIM_R];
#pragma unroll
            for (int j = 0; j < 4 + 2 * SSIM_R; ++j) v[j] = hq[r0 + j][col];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) t = __builtin_fmaf(ssim_w(k), v[o + k], t);
                acc[q][o] = t;
            }
        }
    }
    const int gx = x0 + col;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int gy = y0 + r0 + o;
        if (gx < p.W && gy < p.H) {
            const size_t at = p.at(ch, gy, gx);
            const float x = img1[at], y = img2[at], d = x - y;
            const float sign = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            dL_dimg1[at] = gs * (acc[0][o] + 2.0f * x * acc[1][o] + y * acc[2][o]) + gl * sign;
        }
    }
}

int fail_loss(const char* what)
{
    hgs::set_last_error(what);
    return HGS_ERR_INVALID_ARGUMENT;
}

// number of tiles, and the 1-D grid that covers them in XCD bands (tile_of_workgroup)
int64_t loss_tiles(int C, int H, int W) { return (int64_t)((W + SSIM_TW - 1) / SSIM_TW) * ((H + SSIM_TH - 1) / SSIM_TH) * C; }
dim3 loss_grid(int64_t tiles) { return dim3((unsigned)(((tiles + 7) / 8) * 8)); }

}  // namespace

extern "C" size_t hgs_ssim_l1_workspace(int32_t C, int32_t H, int32_t W)
{
    if (C < 1 || H < 1 || W < 1) return 0;
    return sizeof(float2) * (size_t)loss_tiles(C, H, W);
}

extern "C" int32_t hgs_ssim_l1_forward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, float* maps,
                                       void* workspace, float* out, void* stream)
{
    if (C < 1 || H < 1 || W < 1 || C > 65535) return fail_loss("ssim_l1_forward: need 1 <= C <= 65535, H >= 1, W >= 1");
    if (!img1 || !img2 || !workspace || !out) return fail_loss("ssim_l1_forward: null pointer");
    if (((uintptr_t)workspace & 7) != 0) return fail_loss("ssim_l1_forward: the workspace must be 8-byte aligned");
    const int64_t tiles = loss_tiles(C, H, W);
    if (tiles > (1ll << 30)) return fail_loss("ssim_l1_forward: image too large");
    const dim3 g = loss_grid(tiles);
    const Plane p{C, H, W};
    hipStream_t st = (hipStream_t)stream;
    if (maps) hipLaunchKernelGGL(ssim_l1_forward_kernel<true>, g, dim3(256), 0, st, p, img1, img2, maps, (float2*)workspace);
    else hipLaunchKernelGGL(ssim_l1_forward_kernel<false>, g, dim3(256), 0, st, p, img1, img2, maps, (float2*)workspace);
    hipLaunchKernelGGL(ssim_l1_reduce_kernel, dim3(1), dim3(256), 0, st, (int)tiles, (const float2*)workspace,
                       (double)C * H * W, out);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("ssim_l1_forward: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}

extern "C" int32_t hgs_ssim_l1_backward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* maps,
                                        const float* g_ssim_mean, const float* g_l1_sum, float* dL_dimg1, void* stream)
{
    if (C < 1 || H < 1 || W < 1 || C > 65535) return fail_loss("ssim_l1_backward: need 1 <= C <= 65535, H >= 1, W >= 1");
    if (!img1 || !img2 || !dL_dimg1) return fail_loss("ssim_l1_backward: null pointer");
    if (g_ssim_mean && !maps) return fail_loss("ssim_l1_backward: a gradient of the SSIM term needs forward's maps");
    const int64_t tiles = loss_tiles(C, H, W);
    if (tiles > (1ll << 30)) return fail_loss("ssim_l1_backward: image too large");
    const dim3 g = loss_grid(tiles);
    hipLaunchKernelGGL(ssim_l1_backward_kernel, g, dim3(256), 0, (hipStream_t)stream, Plane{C, H, W}, img1, img2,
                       g_ssim_mean ? maps : nullptr, g_ssim_mean, g_l1_sum, dL_dimg1);
    if (hipGetLastError() != hipSuccess) {
        hgs::set_last_error("ssim_l1_backward: kernel launch failed");
        return HGS_ERR_HIP;
    }
    return HGS_OK;
}
