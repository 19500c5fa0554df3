// Shared declarations for the HIP rasterizer (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hgs_rasterizer.h"

namespace hgs {

constexpr int TILE = 16;        // 16x16 pixel tiles
constexpr int WAVE = 64;        // CDNA wavefront
constexpr float NEAR_Z = 0.2f;  // near cull
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_STOP = 0.0001f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// Sorted list values: low 28 bits = Gaussian index, top 4 bits = quad coverage mask (bit q: the splat may
// reach alpha >= 1/255 somewhere in the tile's 8x8 quad q = qx + 2 qy; conservative, see binning.hip).
constexpr uint32_t GID_BITS = 28;
constexpr uint32_t GID_MASK = (1u << GID_BITS) - 1u;

// Per-Gaussian record written by the preprocess kernel and gathered by the blend kernels through the scalar cache.
// 64 bytes, 64-byte aligned: a record is exactly ONE cache line.  (It was 48 bytes packed; the blend kernels' record
// fetches then touched 1.5 lines on average -- every second record straddles a line boundary -- and the forward blend,
// which waits on the scalar cache a quarter of its time, runs ~10 % faster with one line per record.)
struct alignas(64) Splat {
    float x, y;        // pixel-space mean
    // The blend kernels work in the log2 domain: alpha = exp2(e), one v_exp_f32 with no multiply before or after, and
    // evaluate the exponent through the conic's Cholesky factors,
    //     e = L - (la dx + lb dy)^2 - (lc dy)^2,     L = log2(opacity),
    // la = sqrt(A'), lb = B' / (2 la), lc = sqrt(C' - lb^2) for (A', B', C') = fl(LOG2E * (conic.xx/2, conic.xy, conic.yy/2)):
    // the same five instructions as A dx^2 + B dx dy + C dy^2 + L, but e <= L holds in floating point BY CONSTRUCTION
    // (two squares are subtracted, rounding is monotonic), so the reference's `if (power > 0) continue` can never fire and
    // the blend kernels do not test it -- one compare less per pixel and list entry in both of them.
    float la, lb, lc;
    float log2_opacity;
    float r, g, b;     // view-dependent colour after +0.5 / clamp
    float depth;       // view-space z; its raw bits are the low half of the sort key
    int32_t radius;    // 0 => culled
    uint32_t clamped;  // bit c set => colour channel c was clamped at 0
    // fourth 16-byte quarter (not fetched by the blend kernels): the half-conic itself, (ca, cb, cc) = fl(LOG2E *
    // (-conic.xx/2, -conic.xy, -conic.yy/2)), for emit's coverage masks and the per-Gaussian backward; and L again, so that
    // emit needs three quarters of the record, not four
    float ca, cb, cc;
    float log2_opacity_again;
};
static_assert(sizeof(Splat) == 64, "Splat layout");

struct Camera {  // small by-value kernel argument
    int W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, mod;
    int D, M;
    float scale_grad_factor;  // backward: mod (the true dL/dscale) or 1 (HGS_BWD_UPSTREAM_SCALE_GRAD)
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- scratch layouts (single source of truth, also served by hgs_scratch_offset) ----
constexpr int BIN_GROUP = 1024;           // most Gaussians per binning workgroup (one per thread)
constexpr int BIN_LDS_TILES = 22 * 1024;  // largest tile count whose u32 array fits the 160 KB of LDS next to emit's 68 KB of staging
// (round 6) ... and up to twice that many tiles -- 3840x2160: 32 400 -- the binning kernels keep their per-tile LDS counters in 16-bit
// halves (a group holds at most 1 024 Gaussians, each at most once per tile: binning_walk.h TileHist); emit then adds its counter to the
// segment start + run start it fetches per pair.  Until round 6 such frames counted and emitted on global atomics (a 4K frame: 102 us).
constexpr int BIN_LDS16_TILES = 44 * 1024;
inline int num_tiles_of(int H, int W) { return ((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE); }
// Binning cells: BIN_CELL x BIN_CELL tiles.  On large frames the Gaussians are counting-sorted by the cell of their
// rectangle's first tile, and the binning groups are runs of that order -- neighbours on screen (binning.hip).
constexpr int BIN_CELL = 4;
constexpr uint32_t BIN_SPREAD_MIN = 256;  // splats of more tiles than this are BIG: not sorted by the cell of their first tile (preprocess.hip)
// Round 5: binning groups of their own for the BIG splats of a frame -- a trained scene keeps 1 220 background splats of
// 256 .. 8 160 tiles, 30 % of its pairs; round 4 dealt them to pseudo-random cells, three or four to EVERY group, whose tile window then
// is the whole screen.  Now BIG_PER_GROUP of them per group-sized run of `order` (the other slots empty), spread over the group's waves
// (the count kernel walks a big splat's tiles with the wave that holds it), at most BIG_GROUPS_CAP such groups (the rest fills whole
// groups), and these groups go FIRST in the count and emit grids (group_of_block: at the end they were the launches' tail).
// Trained profile, same box: scatter + count + scan 37.5 -> 32.4-34.1 us, emit + scan 53.6 -> 49.9-51.5, 1 518-1 525 -> 1 540-1 551 FPS
// at 16 / 24 per group; 32: 1 538; 8 overflows the cap (1 242).  Without the reordering the same grouping measured SLOWER than the
// spreading (1 437-1 470 against 1 504: DESIGN_HISTORY.md) -- VERDICT r4's costed -25 us was optimistic by 18.
// HGS_BIG_PER_GROUP: another figure; 0 = round 4's spreading.
constexpr int BIG_PER_GROUP = 24, BIG_GROUPS_CAP = 128;
// binning groups a frame of P Gaussians can need: the runs of `order`, the padded big groups, and one for the rounding between them
inline size_t bin_groups_for(int P, int g) { return (size_t)((P + g - 1) / g) + (size_t)BIG_GROUPS_CAP + 1; }
constexpr int BIN_MAX_CELLS = 2048;  // cells whose populations one scatter workgroup prefix-sums (a 1080p frame has 510)
inline int num_cells_of(int gx, int gy) { return ((gx + BIN_CELL - 1) / BIN_CELL) * ((gy + BIN_CELL - 1) / BIN_CELL); }
enum { BIN_NONE = 0, BIN_IN_ORDER = 1, BIN_BY_CELL = 2 };  // who forms the binning groups (preprocess.hip, binning.hip)
// BIN_BY_CELL costs two more launches (scatter, group count) and saves the per-group passes over all tiles plus most of the
// atomics and partial-line key stores: it pays on frames with many Gaussians AND many tiles.  (Measured: 200k / 1080p +1.4 %,
// 310k / 1080p +2 %, 35k-100k / 1080p +0.3 .. +1.5 %, 20k / 1080p even; 110k at 512x512 -2.7 %, the 6 890-Gaussian SMPL
// template at 512x512 -7.5 %.)
// (round 6) ... and a frame whose Gaussians cover a small part of the screen -- the human-only render of a training step at the capture's
// size: 110 210 Gaussians on 3 064 of 8 160 tiles -- is binned in order whatever its P: its groups' windows are small either way, and
// the two extra launches do not pay (shape scan: +6.5 %).  `nonempty_before`: non-empty tiles of the shape's last frame, -1 = unknown
// (a launch-size hint like the others: lists and images do not depend on the binning mode).
inline int bin_mode_for(int P, int num_tiles, int num_cells, int group, int nonempty_before = -1)
{
    if (!group) return BIN_NONE;
    if (nonempty_before >= 0 && 2 * nonempty_before < num_tiles) return BIN_IN_ORDER;
    return (P >= 32768 && num_tiles >= 4096 && num_cells <= BIN_MAX_CELLS) ? BIN_BY_CELL : BIN_IN_ORDER;
}
// Gaussians per binning group (a multiple of 64, at most BIN_GROUP): the preprocess kernel and emit share this partition
// (a group = a workgroup).  A group is the unit of parallelism of both kernels, and either runs one workgroup per CU, so
// the groups should just fill the 256 CUs in one round: ~250 groups when P allows it (200 000 Gaussians: 241 groups of
// 832, not 196 of 1024 on three quarters of the chip; the 6 890 Gaussians of the SMPL template: 108 groups of 64, not 7
// of 1024).  Every group also pays for passes over all tiles (LDS histogram, flush, cursor load), which bounds how small
// a group may be on a frame with many tiles (~4 M tile visits in total).  0: more tiles than fit LDS -- the
// global-atomics fallback kernels.
inline int bin_group_for(int P, int num_tiles)
{
    if (num_tiles > BIN_LDS16_TILES) return 0;
    // ~250 groups per "round" of the 256 CUs, and whole rounds: 300 000 Gaussians are 469 groups of 640 (two rounds),
    // not 293 of 1024 (one round and a nearly empty second one that takes just as long)
    const long long rounds = ((long long)P + 250 * BIN_GROUP - 1) / (250 * BIN_GROUP);
    const long long want = ((long long)P + 250 * rounds - 1) / (250 * rounds);
    const long long floor_by_tiles = ((long long)P * num_tiles + (4ll << 20) - 1) / (4ll << 20);  // <= 4 Mi tile visits
    long long g = want > floor_by_tiles ? want : floor_by_tiles;
    g = (g + 63) / 64 * 64;
    return (int)(g < 64 ? 64 : (g > BIN_GROUP ? BIN_GROUP : g));
}
struct GeomLayout {
    size_t splats, tiles_touched, cell_slot, order, windows, run_start, total;
    GeomLayout(int P, int num_tiles) {
        size_t o = 0;
        splats = o;         o = align_up(o + sizeof(Splat) * (size_t)P);
        tiles_touched = o;  o = align_up(o + 4 * (size_t)P);
        cell_slot = order = windows = run_start = o;
        if (const int g = bin_group_for(P, num_tiles)) {
            const size_t groups = bin_groups_for(P, g);
            cell_slot = o;  o = align_up(o + 8 * (size_t)P);   // uint2 per Gaussian: its binning cell, its slot inside the cell
            order = o;      o = align_up(o + 4 * groups * (size_t)g);   // the Gaussians that touch a tile, sorted by cell; then the big ones, BIG_PER_GROUP per group-sized run
            // uint4 per binning group: the tile window (x0, y0, width, height) its rectangles span; then one more whose .x
            // is the number of Gaussians in `order`
            windows = o;    o = align_up(o + 16 * (groups + 1));
            // run_start[group][tile]: where, inside the tile's segment, the run of binning group `group` begins -- handed
            // out by the group-count kernel's returning atomics, consumed by emit.  Only (group, tile) pairs with at
            // least one entry are ever written or read.
            run_start = o;  o = align_up(o + 4 * groups * (size_t)num_tiles);
        }
        total = o;
    }
};
constexpr int ACT_PAD = 16;
constexpr int NUM_LISTS = 5;  // four per-quad lists + one "any quad" list per tile
// Depth-segmented backward (blend.hip): on sparse frames the forward blend leaves a per-pixel checkpoint -- (T, colour
// prefix) -- every CKPT_SEG positions of a quad's compacted list, so that a backward wave can start at any of them.
// A checkpoint SLOT holds the 256 pixels of one tile (four quads x 64 lanes x float4 = 4 KB).  Tile t owns the slots
// [seg_first[t], seg_first[t + 1]) with seg_first[t] = (start of the tile's list segment >> CKPT_SHIFT) + t: at least
// ceil(list length / CKPT_SEG) of them, no prefix sum beyond the one the scan already does, (N >> CKPT_SHIFT) + T in all.
constexpr int CKPT_SHIFT = 5, CKPT_SEG = 1 << CKPT_SHIFT;
// On a DENSE frame that was given a checkpoint buffer (the caller expects long tiles), tiles from this many entries on leave
// checkpoints too and go through the depth-segmented backward; the one-wave-per-tile kernel skips them.  A dense frame's
// backward otherwise lasts as long as its deepest tile's chain -- a person in front of a scene (the joint render of HUGS).
// slot_tile = CKPT_SLOT_NONE marks the slots of tiles that left none.
#ifndef HGS_CKPT_DEEP_MIN   // (A/B builds: tools/ab_build.sh name -DHGS_CKPT_DEEP_MIN=...)
#define HGS_CKPT_DEEP_MIN 512u
#endif
#ifndef HGS_DEEP_BWD_MIN
#define HGS_DEEP_BWD_MIN 2048u
#endif
constexpr uint32_t CKPT_DEEP_MIN = HGS_CKPT_DEEP_MIN, CKPT_SLOT_NONE = 0xFFFFFFFFu;
// n_total[3]: which tiles of the frame leave checkpoints (decided by the scan; blend_fwd.h ckpt_leave)
enum : uint32_t { CKPT_KIND_DEEP = 0u, CKPT_KIND_ALL = 1u, CKPT_KIND_NONE = 2u };
// a DENSE frame asks for checkpoints (and the segmented backward of its deep tiles) when its shape's last frame had a list beyond
// this many entries (tile_scan_kernel counts them: FrameHistory::n_deep)
constexpr uint32_t DEEP_BWD_MIN = HGS_DEEP_BWD_MIN;
struct ImageLayout {
    size_t final_T, n_contrib, ranges, act_count, cursor, large_tiles, n_total, seg_first, quad_nproc, total;
    ImageLayout(int H, int W) {
        size_t S = (size_t)H * W, T = (size_t)((H + TILE - 1) / TILE) * ((W + TILE - 1) / TILE);
        size_t o = 0;
        final_T = o;    o = align_up(o + 4 * S);
        n_contrib = o;  o = align_up(o + 4 * S);
        ranges = o;     o = align_up(o + 8 * T);
        act_count = o;  o = align_up(o + 4 * T * NUM_LISTS);   // entries in each tile's compacted lists
        cursor = o;     o = align_up(o + 4 * T);   // start of each tile's segment (the fallback emit path advances it with atomics)
        large_tiles = o; o = align_up(o + 4 * T);  // the frame's LONG tiles (sorted ahead of the fused kernel, blended by its deep workers)
        n_total = o;    o = align_up(o + 64);      // [0] N, [1] capacity-exceeded gate, [2] number of long-tile candidates, [3] sparse-frame flag, [4] long-tile threshold
        seg_first = o;  o = align_up(o + 4 * (T + 1));  // first checkpoint slot of each tile; [T] = number of slots
        quad_nproc = o; o = align_up(o + 16 * T);       // list entries the forward blend walked, per (tile, quad)
        total = o;
    }
};
struct CkptLayout {
    size_t slot_tile, state, total;
    size_t slots;
    // slots a frame of N list entries can need at most: the sparse layout (a dense frame packs its deep tiles' slots: far fewer)
    static size_t slots_for(int64_t N, int num_tiles) { return (size_t)((N < 0 ? 0 : N) >> CKPT_SHIFT) + (size_t)num_tiles; }
    CkptLayout(int64_t N, int num_tiles) : CkptLayout(slots_for(N, num_tiles)) {}
    explicit CkptLayout(size_t slots_) {
        slots = slots_ < 1 ? 1 : slots_;
        size_t o = 0;
        slot_tile = o;  o = align_up(o + 4 * slots);      // which tile a slot belongs to (written by the tile's forward workgroup)
        state = o;      o = align_up(o + 4096 * slots);   // float4 [slot][quad][lane]: (T, C0, C1, C2) before list position (k + 1) * CKPT_SEG
        total = o;
    }
};
struct BinningLayout {
    size_t keys, list, scratch, act, parts, total;
    size_t act_stride;  // entries between consecutive compacted-list arrays
    size_t max_parts;   // 64-byte records of the long-tile plan (binning.hip): a list beyond 4 096 entries is sorted in parts of
                        // 3 072 .. 4 096 entries by several workgroups
    explicit BinningLayout(int64_t N) {
        size_t n = (size_t)(N < 1 ? 1 : N);
        size_t o = 0;
        keys = o;       o = align_up(o + 8 * n);   // bucket-scattered, unsorted sort keys: depth bits << 32 | gaussian << 4 | mask
        list = o;       o = align_up(o + 8 * n);   // sorted: (pos1 << 32) | mask << 28 | gaussian
        scratch = o;    o = align_up(o + 8 * n);   // keys of tiles too long for LDS (fallback path of tile_sort)
        // act: NUM_LISTS arrays with the sorted list's indexing; array q holds, at a tile's offsets, the tile's entries
        // that cover quad q (q == 4: any quad), compacted to the front of the tile's slot.  ACT_PAD entries of slack
        // between arrays and at both ends: the blend kernels prefetch a few entries past either end of a list.
        act_stride = n + ACT_PAD;
        act = o;        o = align_up(o + 8 * (NUM_LISTS * act_stride + 2 * ACT_PAD));
        max_parts = n / 2048 + 64;   // (a part holds > 2 048 entries but for a tile's last one: at most n / 4096 of those)
        parts = o;      o = align_up(o + 64 * max_parts);
        total = o;
    }
};

// kernels / launchers (defined in the .hip files)
void set_last_error(const char* msg);  // hgs_api.hip

// The A/B switches that launch paths consult, read from the environment ONCE (at the first frame) -- not with a getenv per launch;
// hgs_reload_switches() (tests, A/B tools that flip them inside one process) reads them again.  Published as an immutable snapshot
// that every entry point copies at entry: switches() below is that copy, stable for the whole frame (hgs_api.hip).
struct Switches {
    int bin_mode;               // HGS_BIN_MODE: 0 = the library's choice, 'c' = by cell, 'o' = in order
    bool bwd_two_launches;      // HGS_BWD_TWO_LAUNCHES=1: dense frames with checkpoints run the two backward forms as two launches
    bool deep_forward;          // HGS_DEEP_FORWARD=0: long tiles are blended by one wave per quad like any other tile
    int long_min_sparse;        // HGS_LONG_MIN_SPARSE (0: default)
    int long_min_dense;         // HGS_LONG_MIN_DENSE  (0: default; set = applies whatever the frame's deepest list)
    bool emit_scan;             // HGS_EMIT_SCAN=0: always the stand-alone tile scan kernel (else: folded into emit where it applies)
    bool k1_stage_sh;           // HGS_K1_STAGE_SH=1: the preprocess kernel fetches the SH rows through LDS (measured no faster: off)
    int big_per_group;          // HGS_BIG_PER_GROUP: big splats per binning group of their own (default BIG_PER_GROUP; 0: spread over the cells as round 4 did)
    bool bwd_segmented;         // HGS_BWD_SEGMENTED=0: never leave checkpoints (backward then runs one wave per quad / per tile)
    bool fused_sort_blend;      // HGS_FUSED_SORT_BLEND=0: separate tile-sort and forward-blend kernels
    int bwd_waves_per_tile;     // HGS_BWD_WAVES_PER_TILE: 0 = by the frame's kind, 1 / 4 = forced (frames without checkpoints)
    int k8_coop;                // HGS_K8_COOP: -1 = default, else coop_mode of the per-Gaussian backward
    int deep_min;               // HGS_DEEP_MIN: long tiles are blended split by depth only beyond this many entries (0: every long tile)
    int frame_kind;             // HGS_FRAME_KIND: 0 = the scan's rule, 's' = every frame sparse, 'd' = every frame dense (tools/shape_scan.py)
};
const Switches& switches();

// mode = bin_mode_for(): what the kernel does for the binning besides its own work (see preprocess.hip).  `counters`: the
// per-cell (BIN_BY_CELL) or per-tile (BIN_IN_ORDER) counters, ZERO on entry (hgs_api.hip keeps self-cleaning counter
// arrays per stream).
void launch_preprocess(const hgs_forward_args& a, const Camera& cam, Splat* splats, uint32_t* tiles_touched, int mode,
                       uint32_t* counters, uint2* cell_slot, uint32_t* run_start, int group, int big_per_group, hipStream_t st);
void launch_preprocess_backward(const hgs_backward_args& a, const Camera& cam, const Splat* splats, hipStream_t st);
void launch_mark_visible(int P, const float* means3D, const float* V, uint8_t* present, hipStream_t st);

void launch_count(int P, const Camera& cam, const Splat* splats, uint32_t* tile_count, hipStream_t st);
// BIN_BY_CELL, between the preprocess kernel and the tile scan: `order` (the Gaussians sorted by cell), then per
// binning group (`group` consecutive entries of `order`) the tile window, the per-tile pair counts (added to tile_count with
// one returning atomic per touched tile) and the group's run offsets
void launch_spatial_groups(int P, const Camera& cam, const Splat* splats, const uint32_t* cell_count, const uint2* cell_slot,
                           uint32_t* order, uint4* windows, uint32_t* tile_count, uint32_t* run_start, int group, int big_per_group, hipStream_t st);
// (re-zeroes tile_count, and cell_count if given, behind itself)
void launch_tile_scan(uint32_t* tile_count, int num_tiles, uint32_t* cell_count, int num_cells, uint2* ranges, uint32_t* cursor,
                      uint32_t* n_total, uint32_t* large_tiles, uint32_t* seg_first, uint32_t capacity,
                      unsigned long long* host_slot, uint32_t ticket, uint32_t ckpt_cap, hipStream_t st);
void launch_emit(int P, const Camera& cam, const Splat* splats, uint32_t* cursor, const uint32_t* run_start, const uint32_t* order,
                 const uint4* windows, int group, int big_per_group, uint64_t* keys, const uint32_t* gate, hipStream_t st);
// Frames of at most 8 192 tiles whose binning capacity is known up front: emit and the tile scan as ONE launch (binning.hip,
// emit_scan_kernel).  `arrival`: one zero uint32 next to the per-stream counters, self-resetting.
bool emit_scan_applies(int bin_mode, int num_tiles, int group, int P);
void launch_emit_scan(int P, const Camera& cam, const Splat* splats, const uint32_t* run_start, const uint32_t* order, const uint4* windows, int group,
                      int big_per_group, uint64_t* keys, uint32_t* tile_count, uint32_t* cell_count, int num_cells, uint2* ranges, uint32_t* cursor, uint32_t* n_total,
                      uint32_t* large_tiles, uint32_t* seg_first, uint32_t capacity, unsigned long long* host_slot, uint32_t ticket, uint32_t ckpt_cap,
                      uint32_t* arrival, hipStream_t st);
// act points at the first entry of list array 0 (after the front pad)
// fb != nullptr: the small-tile sort kernel also blends its tile (forward), see binning.hip
// Checkpoints of the forward blend for the depth-segmented backward; state == nullptr: none.  They are written only when
// the frame is sparse (*sparse != 0, decided by the tile scan), which is also when backward reads them.
struct Ckpt {
    float4* state; uint32_t* slot_tile; const uint32_t* seg_first; uint32_t* quad_nproc; const uint32_t* sparse;
};
struct FusedBlend {
    Camera cam; uint32_t lastg; const Splat* splats; const float* bg; float* out_color; float* final_T; uint32_t* n_contrib; int clamp_output;
    Ckpt ck;
};
// What the library remembers of a stream's last frame (hgs_api.hip): launch-size hints only -- results never depend on them.
struct FrameHistory {   // -1: unknown
    int32_t n_long = -1, n_huge = -1;   // lists that were long / beyond 4 096 entries
    int32_t n_deep = -1, sparse = -1;   // lists beyond 2 048 entries; whether the frame was sparse
    int32_t n_nonempty = -1;            // tiles with a list
    int32_t deep_blend = -1;            // whether its long tiles were blended split by depth (the scan's n_total[8], 0 without long tiles)
    int64_t wait_ns = -1;               // how long the shape's frames waited for N lately (hgs_api.hip wait_for_slot sleeps through most of a long one)
    int32_t no_ckpt = -1;               // a sparse frame that left no checkpoints (CKPT_KIND_NONE): the next one is not given a buffer
};
void launch_tile_sort(const uint2* ranges, int num_tiles, const uint64_t* keys, uint64_t* list, uint64_t* scratch,
                      uint64_t* act, size_t stride, uint32_t* act_count, const uint32_t* large_tiles,
                      uint32_t* n_total, void* parts, uint32_t max_parts, bool small_tiles, bool long_tiles, const FusedBlend* fb,
                      const FrameHistory& hist, hipStream_t st);

// Workgroups that blend the long tiles' quads by depth (blend_fwd.h): at the FRONT of the grid of the kernel that blends the
// other tiles (a multiple of 8: the tile -> XCD mapping behind them stays what it is without them).
inline uint32_t deep_workers_for(int num_tiles) { const uint32_t w = 4u * (uint32_t)num_tiles; return (w < 2048u ? w : 2048u) + 7u & ~7u; }
// HGS_DEEP_FORWARD=0: long tiles are blended by one wave per quad like any other tile (A/B measurements, parity tests)
inline bool deep_forward_enabled() { return switches().deep_forward; }
void launch_blend_forward(const Camera& cam, int P, const uint2* ranges, const uint64_t* act, size_t act_stride,
                          const uint32_t* act_count, const Splat* splats, const float* bg, float* out_color,
                          float* final_T, uint32_t* n_contrib, const uint32_t* n_total, bool clamp_output,
                          const uint32_t* large_tiles, bool all_tiles, bool long_sorted, const Ckpt& ck, hipStream_t st);
// grad_accum: [P][12] floats, zero on entry: mean2D.x, mean2D.y, conic xx, xy, yy, opacity, r, g, b, pad x3
void launch_blend_backward(const Camera& cam, int P, const uint2* ranges, const uint64_t* act, size_t act_stride,
                           const uint32_t* act_count, bool sparse_frame, const Splat* splats, const float* bg,
                           const float* final_T, const uint32_t* n_contrib, const float* dL_dpix, float* grad_accum,
                           const Ckpt& ck, int64_t num_rendered, int64_t dense_slots, hipStream_t st);   // dense_slots < 0: not known

}  // namespace hgs
