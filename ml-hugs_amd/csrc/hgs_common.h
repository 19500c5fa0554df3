// Shared declarations for the HIP rasterizer (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hgs_rasterizer.h"

namespace hgs {

constexpr int TILE = 16;        // 16x16 pixel tiles
constexpr int WAVE = 64;        // CDNA wavefront
constexpr float NEAR_Z = 0.2f;  // near cull
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_STOP = 0.0001f;

// Sorted list values: low 28 bits = Gaussian index, top 4 bits = quad coverage mask (bit q: the splat may
// reach alpha >= 1/255 somewhere in the tile's 8x8 quad q = qx + 2 qy; conservative, see binning.hip).
constexpr uint32_t GID_BITS = 28;
constexpr uint32_t GID_MASK = (1u << GID_BITS) - 1u;

// Per-Gaussian record written by the preprocess kernel and gathered by the blend kernels.
// 48 bytes, 16-byte aligned: three dwordx4 (scalar or vector) loads.
struct alignas(16) Splat {
    float x, y;        // pixel-space mean
    float ca, cb, cc;  // half-conic (A,B,C) = (-conic.xx/2, -conic.xy, -conic.yy/2): power = A dx^2 + B dx dy + C dy^2
    float opacity;
    float r, g, b;     // view-dependent colour after +0.5 / clamp
    float depth;       // view-space z; its raw bits are the low half of the sort key
    int32_t radius;    // 0 => culled
    uint32_t clamped;  // bit c set => colour channel c was clamped at 0
};
static_assert(sizeof(Splat) == 48, "Splat layout");

struct Camera {  // small by-value kernel argument
    int W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, mod;
    int D, M;
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- scratch layouts (single source of truth, also served by hgs_scratch_offset) ----
struct GeomLayout {
    size_t splats, tiles_touched, offsets, scan_tmp, total;
    explicit GeomLayout(int P) {
        size_t o = 0;
        splats = o;        o = align_up(o + sizeof(Splat) * (size_t)P);
        tiles_touched = o; o = align_up(o + 4 * (size_t)P);
        offsets = o;       o = align_up(o + 4 * (size_t)P);
        scan_tmp = o;      o = align_up(o + 4 * ((size_t)P / 1024 + 2) + 64);
        total = o;
    }
};
struct ImageLayout {
    size_t final_T, n_contrib, ranges, total;
    ImageLayout(int H, int W) {
        size_t S = (size_t)H * W, T = (size_t)((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
        size_t o = 0;
        final_T = o;   o = align_up(o + 4 * S);
        n_contrib = o; o = align_up(o + 4 * S);
        ranges = o;    o = align_up(o + 8 * T);
        total = o;
    }
};
constexpr int ACT_PAD = 16;
constexpr int NUM_BITMAPS = 5;  // four per-quad bitmaps + one "any quad" bitmap over the sorted list
constexpr int SORT_ITEMS = 16;                     // keys per thread per block
constexpr int SORT_THREADS = 256;
constexpr int SORT_TILE = SORT_ITEMS * SORT_THREADS;  // 4096 keys per block
constexpr int SORT_MAX_BINS = 512;                 // up to 9-bit digits
struct BinningLayout {
    size_t keys, keys_alt, values, values_alt, hist, totals, bitmaps, wprefix, scan_tmp, act, total;
    size_t nblocks, bitmap_words;  // bitmap_words = u64 words per quad bitmap
    explicit BinningLayout(int64_t N) {
        size_t n = (size_t)(N < 1 ? 1 : N);
        nblocks = (n + SORT_TILE - 1) / SORT_TILE;
        size_t o = 0;
        keys = o;       o = align_up(o + 8 * n);
        keys_alt = o;   o = align_up(o + 8 * n);
        values = o;     o = align_up(o + 4 * n + 64);  // +64: the blend kernels fetch list entries 4 at a time
        values_alt = o; o = align_up(o + 4 * n + 64);
        hist = o;       o = align_up(o + 4 * (size_t)SORT_MAX_BINS * nblocks);
        totals = o;     o = align_up(o + 4 * (size_t)SORT_MAX_BINS * 8);
        // 4 bitmaps (one per 8x8 quad of a tile) over the sorted list: bit i of bitmap q <=> entry i covers quad q
        bitmap_words = n / 64 + 4;
        bitmaps = o;    o = align_up(o + 8 * NUM_BITMAPS * bitmap_words);
        // wprefix[q * bitmap_words + w] = number of set bits of all earlier words (quads concatenated): the position
        // of word w's first covering entry in the compacted list `act`
        wprefix = o;    o = align_up(o + 4 * NUM_BITMAPS * bitmap_words);
        scan_tmp = o;   o = align_up(o + 4 * (NUM_BITMAPS * bitmap_words / 1024 + 2) + 64);
        // act: for each quad, the covering entries of the sorted list, in list order, as (pos1 << 32 | gaussian);
        // worst case 4 N entries; ACT_PAD dead entries in front (the backward walk reads pairs downwards) and behind
        act = o;        o = align_up(o + 8 * (NUM_BITMAPS * n + 2 * ACT_PAD));
        total = o;
    }
};

// kernels / launchers (defined in the .hip files)
void launch_preprocess(const hgs_forward_args& a, const Camera& cam, Splat* splats, uint32_t* tiles_touched,
                       hipStream_t st);
void launch_preprocess_backward(const hgs_backward_args& a, const Camera& cam, const Splat* splats, hipStream_t st);
void launch_mark_visible(int P, const float* means3D, const float* V, uint8_t* present, hipStream_t st);

void launch_scan_inclusive(const uint32_t* in, uint32_t* out, uint32_t* tmp, int n, hipStream_t st);
void launch_emit_keys(int P, const Camera& cam, const Splat* splats, const uint32_t* offsets, uint64_t* keys,
                      uint32_t* values, hipStream_t st);
// sorts N (key,value) pairs on key bits [0,num_bits); result ends in (keys_a, vals_a).
// returns which buffer the UNSORTED input must be placed in: 0 => (keys_a, vals_a), 1 => (keys_b, vals_b)
int sort_input_buffer(int num_bits);
void launch_sort_pairs(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* hist,
                       uint32_t* totals, int64_t N, int num_bits, hipStream_t st);
// ranges + quad bitmaps + compacted per-quad entry lists (act points at the first real entry, after the front pad)
void launch_tile_ranges(const uint64_t* keys, const uint32_t* values, int64_t N, uint2* ranges, int num_tiles,
                        uint64_t* bitmaps, size_t bitmap_words, uint32_t* wprefix, uint32_t* scan_tmp, uint64_t* act,
                        hipStream_t st);

void launch_blend_forward(const Camera& cam, const uint2* ranges, const uint64_t* act, const uint32_t* wprefix,
                          const uint64_t* bitmaps, size_t bitmap_words, const Splat* splats, const float* bg,
                          float* out_color, float* final_T, uint32_t* n_contrib, hipStream_t st);
// grad_accum: [P][12] floats, zero on entry: mean2D.x, mean2D.y, conic xx, xy, yy, opacity, r, g, b, pad x3
void launch_blend_backward(const Camera& cam, const uint2* ranges, const uint64_t* act, const uint32_t* wprefix,
                           const uint64_t* bitmaps, size_t bitmap_words, const Splat* splats, const float* bg,
                           const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, float* grad_accum,
                           hipStream_t st);

}  // namespace hgs
