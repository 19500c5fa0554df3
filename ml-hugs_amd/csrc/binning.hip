// Tile binning (scan, key emission, sort, tile ranges), restructured for MI355X:
//
//   (count)     per-tile population: done at the tail of the preprocess kernel (preprocess.hip) -- a 1024-Gaussian
//               workgroup counts into an LDS array and takes one RETURNING atomic per touched tile, which is also
//               where its run starts inside the tile's segment (run_start[group][tile]); count_kernel below is the
//               fallback for frames with more tiles than fit an LDS array
//   tile_scan   one workgroup: exclusive scan of the tile counts -> ranges[tile] = [start,end), N = total (published
//               to the host from the kernel), the capacity gate, the list of long tiles; re-zeroes the counters
//   emit        every (Gaussian, tile) pair takes a slot of its tile's segment -- segment start + its group's run start +
//               an LDS counter -- and stores its 64-bit sort key (depth bits << 32 | Gaussian << 4 | quad coverage
//               mask) there                                                                       -- a bucket scatter
//   tile_sort   one workgroup per tile: bitonic sort of the segment in registers (LDS for long tiles) by (fp32 depth
//               bits, Gaussian index) and, while the segment is at hand, the tile's compacted per-quad lists
//
// The result is, for every tile, exactly the order a stable sort of the 64-bit keys (tile << 32 | depth bits)
// produces (ties: ascending Gaussian index) -- bit-identical to the oracle's sorted list -- without ever moving a
// 64-bit key through a multi-pass global radix sort: ~36 B per entry of HBM traffic instead of ~160 B, and 5 kernel
// launches instead of 20.  The slot order inside a segment before sorting is arbitrary (atomics); the per-tile
// sort is on a total order, so the output is deterministic.
#include <cstdlib>
#include <type_traits>

#include "hgs_common.h"
#include "binning_walk.h"
#ifdef HGS_TRACE
namespace hgs { __device__ unsigned long long* g_trace_buf = nullptr; }
extern "C" int hgs_debug_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(hgs::g_trace_buf), &p, sizeof p); }
#define HGS_TRACE_THIS_FILE
#endif
#include "blend_fwd.h"

namespace hgs {

// ---------------------------------------------------------------------------------------------
// wave / block scan helpers (wave = 64 lanes)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// tile_scan: single workgroup; exclusive scan of the per-tile counts.  ranges[t] = (0,0) for an empty tile (as the
// oracle leaves them); cursor[t] = start of the tile's segment; n_total[0] = N; n_total[1] = (N > capacity), the gate
// that makes the kernels of an optimistically launched frame return at once when the binning buffer was guessed
// too small (hgs_api.hip).  N is also published to the host, straight from this kernel, as ONE 64-bit system-scope
// store (sparse bit << 63 | long-tiles bit << 62 | ticket << 32 | N) into a pinned, host-coherent slot the host polls:
// no copy kernel, no event.  A thread scans eight consecutive tiles (two 16-byte loads, 64 bytes of ranges stored), so
// the 8 160 tiles of a 1080p frame are one pass with two barriers.
constexpr int SORT_CAP_SMALL = 2048, SORT_CAP_LARGE = 8192;  // list lengths the register / LDS tile sorts take
constexpr int SORT_CAP_MID = 4096;   // ... and what one workgroup of the long tiles' sort kernel holds (longer lists are split into parts)
// On SPARSE frames (few non-empty tiles, deep lists: a human-only render) the one-workgroup-per-tile kernel below has the CUs
// mostly idle and lasts as long as its longest tile's sort: there, lists from LONG_MIN_SPARSE entries on already go to the
// long-tile kernel (1 024 threads and a bucket sort instead of a 256-thread bitonic network).
// Round 4: the long tiles' forward blend is depth-parallel (blend_fwd.h, deep_forward_worker), so "long" now also means "worth
// splitting by depth": LONG_MIN_SPARSE comes down from 1 024.  Both thresholds are kernel arguments of the scan (defaults
// below; HGS_LONG_MIN_SPARSE / HGS_LONG_MIN_DENSE override them for A/B measurements); what a frame uses is n_total[4].
constexpr int LONG_MIN_SPARSE = 256;         // sparse frames with DEEP lists (mean non-empty list >= DEEP_MEAN_MIN entries): a 110k-Gaussian human, the SMPL template
constexpr int LONG_MIN_SPARSE_SHALLOW = 1024;  // other sparse frames (what round 3 used for all of them)
// 384 until the end of round 5, which left the 6 890-Gaussian template (mean list 260, composited depth <= 430) on the shallow side.
// Measured again with round 5's forward (A/B builds, two boxes): its sort + forward kernels 48.3-49.6 -> 44.7-45.8 us with 200 or 128,
// unchanged with 280; frames of 3 000-60 000 uniform Gaussians at 256x256 ... 540x960 are deep under either value (tools/ab_build.sh
// name -DHGS_DEEP_MEAN_MIN=...).
#ifndef HGS_DEEP_MEAN_MIN
#define HGS_DEEP_MEAN_MIN 200
#endif
constexpr uint32_t DEEP_MEAN_MIN = HGS_DEEP_MEAN_MIN;
// dense frames: 768 WHEN the frame holds a list beyond 2 048 entries (else 2 048: tile_scan_kernel, dense_min); until round 4: 2 048,
// what the one-workgroup-per-tile sort holds.  A person in front of a scene puts
// hundreds of tiles between 1 024 and 2 048 entries, whose one-wave walks were the tail of the fused kernel: the all-rows step's
// joint render 237 -> 192 us.  768 against 1 024 (same box, kernel trace): the trained-scene profile's fused kernel 223 -> 205 us
// for +1 us of long-tile sort, C4 and the step unchanged; 512 / 384 give the sort back what the blend gains (31 / 43 us).  ("Has long tiles" is also what makes the bindings offer a checkpoint buffer; whether a DENSE
// frame uses it is decided by the library from its own history -- hgs_api.hip, FrameHistory::n_deep -- as before: lists beyond
// 2 048 entries.  C4's joint render, deepest tile 1 900 entries and a throughput- not chain-bound backward, paid 78 us for the
// segmented backward of its 512+-entry tiles when the lower threshold switched the buffer on.)
// (Measured and dropped with it: the tiles between 512 entries and the threshold taken FIRST by workgroups behind the workers --
// a per-workgroup trace had shown such tiles start after 110 us of a 190 us kernel and finish alone --: the all-rows step's
// joint render unchanged (136 against 137 us per render), C2 168.6 -> 171.3, C4 140.2 -> 145.4 us.)
constexpr int LONG_MIN_DENSE = 768;
constexpr uint32_t LONG_MIN_SPARSE_TILES = 16;  // ... when the frame has at least this many of them (a launch has to pay for itself)
// ... and at most this many: what ONE round of the long tiles' kernel takes (its grid, launch_tile_sort).  A sparse frame's threshold is
// the lowest of long_min_sparse (deep lists only) / LONG_MIN_SPARSE_SHALLOW / SORT_CAP_SMALL that leaves no more lists than this.  The
// thresholds were tuned on human-only renders -- a few hundred non-empty tiles --, but "sparse" is every frame under 4 096 non-empty
// tiles: a 720p scene render (3 600 tiles, 100 000 Gaussians, mean list 600) sent 3 000 lists through the 512-workgroup kernel, round
// after round, in front of its fused kernel: sort + forward 214 us against 107 with the threshold at 1 024; 20 000 Gaussians at
// 540 x 960: 125 against 60 us (end of round 5, `HGS_LONG_MIN_SPARSE` sweep; C3's frames have 276 / 346 such lists and keep 256).
constexpr uint32_t LONG_ONE_ROUND = 512;
constexpr uint32_t DEEP_EVEN_TILES = 1000, DEEP_EVEN_L_X10 = 18;   // sparse frames this full and this even blend their long lists one wave per quad (tile_scan_body)
constexpr uint32_t DENSE_LONG_MANY = 832;   // a dense frame with more lists than this beyond long_min_dense takes LONG_MIN_SPARSE_SHALLOW (tile_scan_body)

constexpr int SCAN_ITEMS = 8;  // consecutive tiles per thread and pass: 8 192 tiles per pass of the 1024 threads
constexpr uint32_t N_TOO_MANY = 0xFFFFFFF0u;  // pair counts from here on are reported as "too many" (32-bit list positions)

// Everything the scan reads and writes (a kernel argument of tile_scan_kernel, and of emit_scan_kernel whose extra workgroup runs it)
struct ScanArgs {
    uint32_t* tile_count; int num_tiles; uint32_t* cell_count; int num_cells;
    uint2* ranges; uint32_t* cursor; uint32_t* n_total; uint32_t* large_tiles; uint32_t* seg_first; uint32_t capacity;
    unsigned long long* host_slot; uint32_t ticket, long_min_sparse, long_min_dense_arg;
    uint32_t ckpt_cap;   // checkpoint slots the frame's buffer was laid out for (0: enough for whatever this frame needs)
    uint32_t force_kind; // HGS_FRAME_KIND: 0 = the rule below, 1 = sparse, 2 = dense (A/B measurements: tools/shape_scan.py)
    uint32_t deep_min;   // HGS_DEEP_MIN: long tiles are blended split by depth only beyond this many entries (0: every long tile)
};

// SPARSE or DENSE: the one decision the rest of the frame's path selection hangs on (the backward's form, the checkpoint layout, the
// long-list thresholds) -- taken from what the scan knows of the FRAME, not from its size:
//   n_nonempty          tiles with a list: the waves the one-wave-per-tile backward runs (1 024 SIMDs: four each from 4 096 tiles on)
//   E = sum len^2 / N   the size-biased mean list length: the length of the list a random ENTRY sits in -- equal to the mean on a
//                       uniform frame, several times the mean where a person stands in front of a scene (the deep lists hold most
//                       of the entries, and one wave per tile walks each of them alone)
// Dense = one backward wave per tile (+ the checkpointed walk for its deep tiles, where it holds lists beyond 2 048 entries): fewer
// instructions per entry -- one reduction per (tile, entry) -- but a chain per tile.  From 4 096 non-empty tiles on it keeps the SIMDs
// busy whatever the lists (the rule of rounds 2-5, which called every smaller frame sparse); below that it still wins while the lists
// are short enough for the tile chains to end together: E <= 0.45 (n_nonempty - 800), at most 1 200 -- fitted on the shape scan
// (tools/shape_scan.py, profiles/r6*_shape_scan*.json: 2 040 non-empty tiles: dense up to E ~ 580, 3 600: up to ~1 200; a covered
// 1280x720 frame of 100 000 Gaussians, E = 445: 0.370 -> 0.339 ms; a trained one, E = 638: 0.489 -> 0.395), never under 1 536 tiles
// (a 512x512 frame, any human-only render: the depth-segmented backward from the forward's checkpoints wins at every depth).
constexpr uint32_t DENSE_ALWAYS_TILES = 4096, DENSE_MIN_TILES = 1536, DENSE_E_ORIGIN = 800, DENSE_E_MAX = 1200, DENSE_E_FLAT_MAX = 1600, DENSE_ALWAYS_E_MAX = 830, DENSE_ALWAYS_E_RISE = 680, DENSE_ALWAYS_TAIL_X10 = 26, NO_CKPT_MIN_TILES = 7168;
__device__ __forceinline__ uint32_t frame_is_sparse(uint32_t n_nonempty, unsigned long long total, unsigned long long sum_sq, uint32_t longest, uint32_t force_kind)
{
    if (force_kind) return force_kind == 1u ? 1u : 0u;
    // (round 6) 4 096 tiles and more: dense unless the lists are DEEP -- E beyond DENSE_ALWAYS_E_MAX: 2 097 152 Gaussians of a trained scene
    // at 1080p and above (E = 851 .. 953), a 524 288-Gaussian person at 1080p (E = 1 376).  The wave that walks a whole tile goes as far back
    // as the LAST of its 256 pixels composited, a wave per quad as far as the last of its 64, and in deep lists most quads are done long
    // before their tile is: one wave per quad 8-16 % faster there, 10-45 % slower on every shallower frame of the scan (E <= 705).
    // (the bound rises towards fewer tiles -- + DENSE_ALWAYS_E_RISE from 8 192 tiles down to 4 096: a 1600x900 frame, 5 700 tiles, is 8-22 %
    //  faster dense at E = 785 where 8 160 tiles break even, and 5 % at E = 1 120; 800 000 Gaussians of a trained scene at 1024x1024 and
    //  1366x768 -- 4 096 / 4 128 tiles, E = 1 191 / 1 163, frames of a second holdout grid the first fit (a rise of 340) had not seen --
    //  19-24 % faster dense than one wave per quad without checkpoints: the rise is 680, i.e. 1 510 at 4 096 tiles, 1 240 at 5 700; and
    //  DENSE_ALWAYS_E_MAX itself is 830, not the first fit's 760: 1 500 000 Gaussians of a trained scene at 1080p, E = 812, are 11 % faster
    //  dense, 2 097 152 at 2048x1152, E = 851, 23 % faster a wave per quad)
    // ... and only where the depth is the FRAME's, not a tail's: E <= DENSE_ALWAYS_TAIL_X10 / 10 = 2.6 x the mean list.  A person on a body surface in
    // front of a covered 1080p scene (tools/bench_step.py's joint render: mean 392, E = 2 141, 286 lists beyond 2 048 entries, the longest
    // 10 751) is one wave per tile over 7 800 shallow lists plus the checkpointed walk of the few deep ones -- a dense frame with deep
    // tiles, 6-21 % faster than a wave per quad everywhere (the person grid of the scan, E / mean 2.8 .. 21: the 1600x900 frames at 2.8 by
    // 7-9 %); the frames one wave per quad wins hold E / mean <= 2.35 (a 524 288-Gaussian person alone at 1080p; the trained 2 M scenes 1.27).
    if (n_nonempty >= DENSE_ALWAYS_TILES) {
        const unsigned long long rise = n_nonempty < 2u * DENSE_ALWAYS_TILES ? (unsigned long long)(2u * DENSE_ALWAYS_TILES - n_nonempty) * DENSE_ALWAYS_E_RISE / DENSE_ALWAYS_TILES : 0ull;
        if (sum_sq <= ((unsigned long long)DENSE_ALWAYS_E_MAX + rise) * total) return 0u;
        // (... nor a FLAT frame's, as under 4 096 tiles: the longest list within a quarter of E -- 7-pixel splats covering a 1080p frame, every
        //  list 840-1 130 entries: 10-14 % faster dense; what a wave per quad wins on is the trained scenes' spread, longest ~ 3.5 E)
        if (4ull * longest * total <= 5ull * sum_sq && sum_sq <= (unsigned long long)DENSE_E_FLAT_MAX * total) return 0u;
        return 10.0f * (float)sum_sq * (float)n_nonempty <= (float)DENSE_ALWAYS_TAIL_X10 * (float)total * (float)total ? 1u : 0u;
    }
    if (n_nonempty < DENSE_MIN_TILES) return 1u;
    unsigned long long e_max = min((unsigned long long)DENSE_E_MAX, (unsigned long long)(n_nonempty - DENSE_E_ORIGIN) * 9ull / 20ull);
    // (a FLAT frame -- its longest list within a quarter of E: a covered frame of uniform depth -- has no tile chain that outlasts the
    //  others at any depth the rule's slope allows: up to DENSE_E_FLAT_MAX; 2 097 152 Gaussians at 1280x720, E = 1 485: 7 % faster dense)
    if (e_max == DENSE_E_MAX && 4ull * longest * total <= 5ull * sum_sq) e_max = DENSE_E_FLAT_MAX;
    if (sum_sq > e_max * total) return 1u;   // E = sum_sq / total > e_max
    // ... and only while the frame has no heavy TAIL: E <= 2.5 x the mean list (a person 110 210 Gaussians strong in front of an empty
    // background at 1080p -- 3 064 non-empty tiles, mean 301, E = 855: the tile chains of the body end long after the rest; the
    // checkpointed walk is 19 % faster there -- against a covered 1280x720 frame at E = 988 = its mean, 12 % faster dense)
    const float mean = (float)total / (float)n_nonempty;
    return (float)sum_sq > 2.5f * mean * (float)total ? 1u : 0u;
}

// The scan as a workgroup of 1024 threads.  ZERO: re-zero the counters it has read (the stand-alone kernel, their only reader);
// !ZERO: other workgroups of the same launch read them too, and the last of them to arrive zeroes them (emit_scan_kernel).
template <bool ZERO>
__device__ __forceinline__ void tile_scan_body(const ScanArgs& sa)
{
    uint32_t* __restrict__ tile_count = sa.tile_count;
    const int num_tiles = sa.num_tiles, num_cells = sa.num_cells;
    uint32_t* __restrict__ cell_count = sa.cell_count;
    uint2* __restrict__ ranges = sa.ranges;
    uint32_t* __restrict__ cursor = sa.cursor;
    uint32_t* __restrict__ n_total = sa.n_total;
    uint32_t* __restrict__ large_tiles = sa.large_tiles;
    uint32_t* __restrict__ seg_first = sa.seg_first;
    const uint32_t capacity = sa.capacity, ticket = sa.ticket, long_min_dense_arg = sa.long_min_dense_arg;
    unsigned long long* __restrict__ host_slot = sa.host_slot;
    // (bit 31: the threshold was given explicitly -- HGS_LONG_MIN_DENSE / HGS_LONG_MIN_SPARSE -- and applies whatever the frame's
    //  deepest list / however many lists it makes long)
    const uint32_t long_min_dense = long_min_dense_arg & 0x7FFFFFFFu, dense_unconditional = long_min_dense_arg >> 31;
    const uint32_t long_min_sparse = sa.long_min_sparse & 0x7FFFFFFFu, sparse_unconditional = sa.long_min_sparse >> 31;
    __shared__ uint32_t n_long_sh;
    __shared__ uint32_t wsum[16], wsum2[16];
    __shared__ uint32_t n_large_sparse, n_large_shallow, n_large_dense, n_nonempty, n_huge;
    __shared__ unsigned long long total64;  // the pair count again, in 64 bits: the 32-bit scan wraps silently beyond 2^32
    __shared__ unsigned long long sumsq64;  // sum of the squared list lengths (frame_is_sparse)
    __shared__ uint32_t longest;   // the frame's longest list
    if (threadIdx.x == 0) n_large_sparse = 0, n_large_shallow = 0, n_large_dense = 0, n_nonempty = 0, n_huge = 0, total64 = 0ull, sumsq64 = 0ull, longest = 0u;
    // the cell counters of the counting sort (their readers ran before this kernel) are self-cleaning too
    for (int c = threadIdx.x; c < num_cells; c += 1024) cell_count[c] = 0u;   // (whatever ZERO: emit's workgroups do not read them)
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t carry = 0, carry2 = 0;
    uint32_t my_huge = 0, my_sparse = 0, my_shallow = 0, my_dense = 0;   // this thread's lists beyond each threshold
    // (a frame of at most 8 192 tiles is ONE trip of this loop: the number of non-empty tiles -- dense frame or sparse -- is then
    //  complete behind the trip's barrier, before the checkpoint slots are dealt)
    const bool single_trip = num_tiles <= 1024 * SCAN_ITEMS;
    for (int base = 0; base < num_tiles; base += 1024 * SCAN_ITEMS) {
        const int t0 = base + threadIdx.x * SCAN_ITEMS;
        uint32_t c[SCAN_ITEMS];
        if (t0 + SCAN_ITEMS <= num_tiles) {  // (the counter array is 16-byte aligned and t0 a multiple of 8)
            const uint4 lo = reinterpret_cast<const uint4*>(tile_count + t0)[0], hi = reinterpret_cast<const uint4*>(tile_count + t0)[1];
            c[0] = lo.x, c[1] = lo.y, c[2] = lo.z, c[3] = lo.w, c[4] = hi.x, c[5] = hi.y, c[6] = hi.z, c[7] = hi.w;
            // the counters are self-cleaning: zero again for the next frame on this stream
            if (ZERO) {
                reinterpret_cast<uint4*>(tile_count + t0)[0] = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<uint4*>(tile_count + t0)[1] = make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) {
                c[k] = t0 + k < num_tiles ? tile_count[t0 + k] : 0u;
                if (ZERO && t0 + k < num_tiles) tile_count[t0 + k] = 0u;
            }
        }
        uint32_t mine = 0, nonempty = 0;
        unsigned long long sq = 0ull;   // (a count is below 2^26: eight squares fit easily)
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) mine += c[k], nonempty += c[k] ? 1u : 0u, sq += (unsigned long long)c[k] * c[k];
        const uint32_t inc = wave_inclusive_scan(mine);
        if (lane == 63) wsum[w] = inc;
        // checkpoint slots of a DENSE frame: its deep tiles alone get them, ceil(length / CKPT_SEG) each, packed (see below)
        uint32_t need[SCAN_ITEMS], mine2 = 0, inc2 = 0;
        if (seg_first) {   // (uniform)
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) need[k] = c[k] >= CKPT_DEEP_MIN ? (c[k] + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT : 0u, mine2 += need[k];
            inc2 = wave_inclusive_scan(mine2);
            if (lane == 63) wsum2[w] = inc2;
        }
        // non-empty tiles of the wave (for the sparse-frame decision)
        uint32_t ne = nonempty;
        unsigned long long m64 = mine;  // (eight counts below 2^26 each: `mine` itself cannot wrap)
        uint32_t mx = 0;   // the wave's longest list (the frame's is complete behind the trip's barrier, like the sums)
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) mx = max(mx, c[k]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            ne += (uint32_t)__shfl_xor((int)ne, d, 64);
            m64 += (unsigned long long)__shfl_xor((long long)m64, d, 64);
            sq += (unsigned long long)__shfl_xor((long long)sq, d, 64);
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
        }
        if (lane == 0 && ne) atomicAdd(&n_nonempty, ne), atomicAdd(&total64, m64), atomicAdd(&sumsq64, sq), atomicMax(&longest, mx);
        __syncthreads();
        uint32_t before = 0, total = 0, before2 = 0, total2 = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t v = wsum[k];
            if (k < w) before += v;
            total += v;
        }
        if (seg_first) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = wsum2[k];
                if (k < w) before2 += v;
                total2 += v;
            }
        }
        const bool dense_now = single_trip && !frame_is_sparse(n_nonempty, total64, sumsq64, longest, sa.force_kind);   // (the packed slot layout can be written right here)
        __syncthreads();
        uint32_t start = carry + before + inc - mine;
        uint32_t st[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            st[k] = start;
            start += c[k];
        }
        if (t0 + SCAN_ITEMS <= num_tiles) {
            uint4* r4 = reinterpret_cast<uint4*>(ranges + t0);
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k += 2)
                r4[k / 2] = make_uint4(c[k] ? st[k] : 0u, c[k] ? st[k] + c[k] : 0u, c[k + 1] ? st[k + 1] : 0u,
                                       c[k + 1] ? st[k + 1] + c[k + 1] : 0u);
            reinterpret_cast<uint4*>(cursor + t0)[0] = make_uint4(st[0], st[1], st[2], st[3]);
            reinterpret_cast<uint4*>(cursor + t0)[1] = make_uint4(st[4], st[5], st[6], st[7]);
            if (seg_first && dense_now) {
                uint32_t at = before2 + inc2 - mine2, sf[SCAN_ITEMS];
#pragma unroll
                for (int k = 0; k < SCAN_ITEMS; ++k) sf[k] = at, at += need[k];
                reinterpret_cast<uint4*>(seg_first + t0)[0] = make_uint4(sf[0], sf[1], sf[2], sf[3]);
                reinterpret_cast<uint4*>(seg_first + t0)[1] = make_uint4(sf[4], sf[5], sf[6], sf[7]);
            } else if (seg_first) {  // first checkpoint slot of each tile (hgs_common.h, CKPT_*)
                const uint32_t b = (uint32_t)t0;
                reinterpret_cast<uint4*>(seg_first + t0)[0] = make_uint4((st[0] >> CKPT_SHIFT) + b, (st[1] >> CKPT_SHIFT) + b + 1u,
                                                                         (st[2] >> CKPT_SHIFT) + b + 2u, (st[3] >> CKPT_SHIFT) + b + 3u);
                reinterpret_cast<uint4*>(seg_first + t0)[1] = make_uint4((st[4] >> CKPT_SHIFT) + b + 4u, (st[5] >> CKPT_SHIFT) + b + 5u,
                                                                         (st[6] >> CKPT_SHIFT) + b + 6u, (st[7] >> CKPT_SHIFT) + b + 7u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k)
                if (t0 + k < num_tiles) {
                    ranges[t0 + k] = c[k] ? make_uint2(st[k], st[k] + c[k]) : make_uint2(0u, 0u);
                    cursor[t0 + k] = st[k];
                    if (seg_first) seg_first[t0 + k] = (st[k] >> CKPT_SHIFT) + (uint32_t)(t0 + k);
                }
            if (seg_first && dense_now) {   // (the ragged tail of the packed layout)
                uint32_t at = before2 + inc2 - mine2;
#pragma unroll
                for (int k = 0; k < SCAN_ITEMS; ++k) {
                    if (t0 + k < num_tiles) seg_first[t0 + k] = at;
                    at += need[k];
                }
            }
        }
        carry += total, carry2 += total2;
    }
    // Which lists ARE long depends on what kind of frame this is, known only now.  Every thread takes the decision for itself from
    // the workgroup's counters (no broadcast, no barrier for it); a frame WITH long lists then collects them into the list the
    // long tiles' kernels and the deep workers walk (large_tiles, n_total[2]); one without (the bench workload) pays four adds
    // per tile and a wave reduction for all this.
    // (round 5) ... and a frame WITHOUT long lists -- the bench workload, every point of the sweep -- does not count at all: one
    // running maximum per tile in the pass above, one wave reduction, and the counting pass below is skipped when the frame's
    // longest list is under the lowest threshold a frame of its kind can choose (round 4 counted for all four thresholds on every
    // frame: +1.2-1.5 us of this one-workgroup kernel).
    __syncthreads();
    const uint32_t sparse_kind = frame_is_sparse(n_nonempty, total64, sumsq64, longest, sa.force_kind);
    const bool force_kind_set = sa.force_kind != 0u;   // (HGS_FRAME_KIND: the forced kinds keep their checkpoints, as the A/B tools expect)
    const uint32_t lowest = sparse_kind ? min(long_min_sparse, (uint32_t)LONG_MIN_SPARSE_SHALLOW) : long_min_dense;
    if (longest > lowest) {   // (workgroup-uniform) count the lists beyond each threshold from the ranges this workgroup wrote
        for (int t00 = 0; t00 < num_tiles; t00 += 8 * 1024) {
            uint2 rg8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + u * 1024 + (int)threadIdx.x;
                rg8[u] = t < num_tiles ? ranges[t] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t len = rg8[u].y - rg8[u].x;
                my_huge += (len > (uint32_t)SORT_CAP_MID ? 1u : 0u) + (len > DEEP_BWD_MIN ? 0x10000u : 0u);   // (two 16-bit counts)
                my_sparse += len > long_min_sparse ? 1u : 0u;
                my_shallow += len > (uint32_t)LONG_MIN_SPARSE_SHALLOW ? 1u : 0u, my_dense += len > long_min_dense ? 1u : 0u;
            }
        }
        uint32_t v[4] = {my_huge, my_sparse, my_shallow, my_dense};
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // wave sum in lane 63: DPP row shifts + row broadcasts
            uint32_t x = v[q];
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
            v[q] = x;
        }
        if (lane == 63) {
            if (v[0]) atomicAdd(&n_huge, v[0]);
            if (v[1]) atomicAdd(&n_large_sparse, v[1]);
            if (v[2]) atomicAdd(&n_large_shallow, v[2]);
            if (v[3]) atomicAdd(&n_large_dense, v[3]);
        }
    }
    if (threadIdx.x == 0) n_long_sh = 0u;
    __syncthreads();
    // more pairs than 32-bit positions can address (N_TOO_MANY and above are reserved): the gate closes whatever the
    // capacity, and the host is told N = 0xFFFFFFFF, which it turns into HGS_ERR_OVERFLOW
    if (total64 >= (unsigned long long)N_TOO_MANY) carry = 0xFFFFFFFFu;
    // a SPARSE frame: one wave per non-empty tile would leave the SIMDs (1 024 of them) under four waves each --
    // the backward blend then splits long tiles over four waves
    const uint32_t sparse = sparse_kind;
    // this frame's long-tile threshold (n_total[4]).  Sparse frames: lists from long_min_sparse entries on are long (sorted
    // ahead, blended split by depth) when the frame's lists are deep on average -- depth parallelism pays for its compose +
    // re-walk overhead only where one wave per quad would walk hundreds of entries -- and from LONG_MIN_SPARSE_SHALLOW on
    // otherwise; in either case only if enough of them exist
    const bool deep_lists = sparse && n_nonempty && total64 >= (unsigned long long)DEEP_MEAN_MIN * n_nonempty;
    const uint32_t huge = n_huge & 0xFFFFu, very_deep = n_huge >> 16;   // lists beyond SORT_CAP_MID / beyond SORT_CAP_SMALL entries
    // (a shallow sparse frame's threshold is the one n_large_shallow counted with; an explicit HGS_LONG_MIN_DENSE below it still applies)
    const uint32_t shallow_min = dense_unconditional ? min((uint32_t)LONG_MIN_SPARSE_SHALLOW, long_min_dense) : (uint32_t)LONG_MIN_SPARSE_SHALLOW;
    // ... the lowest threshold that leaves the long tiles' kernel ONE round of lists (LONG_ONE_ROUND); beyond SORT_CAP_SMALL entries a
    // list is long however many there are (nothing else sorts it)
    uint32_t sparse_min = 0u, n_sparse_long = 0u;
    if (sparse) {
        if (deep_lists && n_large_sparse >= LONG_MIN_SPARSE_TILES && (n_large_sparse <= LONG_ONE_ROUND || sparse_unconditional))
            sparse_min = long_min_sparse, n_sparse_long = n_large_sparse;
        else if (n_large_shallow >= LONG_MIN_SPARSE_TILES && n_large_shallow <= LONG_ONE_ROUND)
            sparse_min = shallow_min, n_sparse_long = n_large_shallow;
        else if (n_large_shallow > LONG_ONE_ROUND) {
            // More lists beyond 1 024 entries than one round of the long tiles' kernel takes.  (round 6) FLAT ones -- the frame's longest
            // list fits one workgroup of that kernel: a covered small frame, 300 000 Gaussians at 512x512, all 1 024 lists between 1 800
            // and 2 500 entries -- are long all the same, from 1 024 entries on: the fused kernel sorts a list of 1 025 .. 2 048 entries
            // with its bitonic network (no room for the buckets), the long tiles' kernel with its bucket sort, and the shape scan has
            // the latter ahead however many lists there are -- but they are NOT blended split by depth (many_flat_long below: the
            // per-quad waves of a thousand tiles fill the SIMDs by themselves; the workers' compose + re-walk is twice the
            // instructions): sort + forward 144 -> 102 us.  With really deep lists among them (a trained scene: the longest 10 337)
            // the threshold stays at 2 048 and the lists beyond it go to the workers, as before (1 024 there: +13-15 %).
            if (longest <= (uint32_t)SORT_CAP_MID) sparse_min = shallow_min, n_sparse_long = n_large_shallow;
            else sparse_min = (uint32_t)SORT_CAP_SMALL, n_sparse_long = very_deep;
        }
    }
    const bool many_flat_long = sparse && n_sparse_long > LONG_ONE_ROUND && longest <= (uint32_t)SORT_CAP_MID && !sparse_unconditional;
    const bool use_sparse = sparse_min != 0u;
    // A dense frame takes the long-tile path (sorted ahead, blended by depth) from long_min_dense entries on only when it holds a list
    // the one-workgroup-per-tile kernel cannot take or walks as a tail (beyond SORT_CAP_SMALL entries); a frame whose deepest lists
    // are merely long (C4's joint render: 1 900) is throughput-bound in that kernel, and the detour cost it 40 us (mid sort 27 + a
    // slower fused kernel, `profiles/r3j_c4_serial_timeline.txt` against round 4's first collection)
    const bool dense_long = very_deep != 0u || dense_unconditional != 0u;
    // (round 6) ... and from LONG_MIN_SPARSE_SHALLOW on where more lists than DENSE_LONG_MANY lie beyond long_min_dense (a second round of
    // the long tiles' kernel in front of the fused one): trained frames with 885 .. 1 262 such lists +0.5 .. 2.7 %, with 265 .. 694 of them
    // -0.2 .. -5.4 %; a person in front of a 600 000-Gaussian scene at 1280x720 (mean list 838: most of 3 600 lists) sort + forward 283 -> 216 us
    const bool dense_many = dense_long && !dense_unconditional && n_large_dense > DENSE_LONG_MANY && long_min_dense < (uint32_t)LONG_MIN_SPARSE_SHALLOW;
    const uint32_t dense_min = dense_many ? (uint32_t)LONG_MIN_SPARSE_SHALLOW : dense_long ? long_min_dense : max(long_min_dense, (uint32_t)SORT_CAP_SMALL);
    const uint32_t threshold = use_sparse ? sparse_min : dense_min;
    const uint32_t any_long = (use_sparse ? n_sparse_long : (dense_many ? n_large_shallow : dense_long ? n_large_dense : 0u)) ? 1u : 0u;
    // (round 6, last scan) ... nor on a sparse frame that already fills the SIMDs with its own quads and has no list that outlasts the
    // others: >= DEEP_EVEN_TILES non-empty tiles (4 000 quad waves for 1 024 SIMDs) and the longest list within DEEP_EVEN_L_X10 / 10 = 1.8 x E.
    // That is a person filling a 512x512 frame (3 units away instead of the canonical rig's 5: 1 020 of 1 024 tiles, longest / E = 1.7) or
    // standing alone in a 1280x720 one: 5-11 % faster one wave per quad on every such point of the scans; the canonical rig (346-739 tiles,
    // 1.9), every trained frame (2.5-3.5) and every person in front of a scene (3-10) keep the workers, which are worth 10-57 % there;
    // uniform frames (1.1) do not care (+-0.6 %).
    const bool even_and_full = sparse && n_nonempty >= DEEP_EVEN_TILES && !sparse_unconditional && sa.deep_min == 0u &&
                               10ull * longest * total64 <= (unsigned long long)DEEP_EVEN_L_X10 * sumsq64;
    const uint32_t deep_flag = ((!sparse || deep_lists) && !many_flat_long && !even_and_full) ? 1u : 0u;
    // Which tiles leave checkpoints for the backward (hgs_common.h, CKPT_KIND_*): every tile of a sparse frame -- except (round 6) on a
    // sparse frame of NO_CKPT_MIN_TILES non-empty tiles and more WITHOUT a heavy tail (E < 1.6 x the mean list: the trained 2 097 152-Gaussian
    // scenes at 1080p and above): 28 000 quad waves and more fill the SIMDs from the end of the lists, and 800 MB of checkpoints cost more
    // than the segmented walk saves (6-16 % of the frame); with a tail (a person in front of an empty background at 1080p) the segmented
    // walk stays, and so it does under NO_CKPT_MIN_TILES (5 700 tiles, 1 500 000 Gaussians: 3 % faster WITH checkpoints; 4 096 tiles: 32 %).
    const bool sparse_no_ckpt = sparse && !force_kind_set && n_nonempty >= NO_CKPT_MIN_TILES &&
                                (float)sumsq64 * (float)n_nonempty < 1.6f * (float)total64 * (float)total64;
    const uint32_t ckpt_kind = !sparse ? (uint32_t)CKPT_KIND_DEEP : sparse_no_ckpt ? (uint32_t)CKPT_KIND_NONE : (uint32_t)CKPT_KIND_ALL;
    if (threadIdx.x == 0) {
        n_total[0] = carry, n_total[1] = carry > capacity || carry == 0xFFFFFFFFu ? 1u : 0u;
        n_total[3] = ckpt_kind, n_total[4] = threshold;
        // [5] parts of the lists beyond SORT_CAP_MID entries (long_tile_plan_kernel appends), [6] how many such lists there are,
        // [7] lists the plan leaves to the one-workgroup fallback
        n_total[5] = 0u, n_total[6] = huge, n_total[7] = 0u;
        // [8]: the long tiles' quads are blended split by depth (the deep workers of the fused kernel) -- on dense frames, and on
        // sparse frames with deep lists; on a shallow sparse frame (the SMPL template: 25 lists beyond 1 024 entries, composited
        // depth <= 430) one wave per quad does as well and the workers' workgroups only stand in the way (measured: +2 us)
        n_total[8] = deep_flag ? max(threshold, sa.deep_min) : 0u;
        if (seg_first) seg_first[num_tiles] = (carry >> CKPT_SHIFT) + (uint32_t)num_tiles;
    }
    // A DENSE frame leaves checkpoints only on its deep tiles (CKPT_DEEP_MIN entries and more): they alone get slots --
    // ceil(length / CKPT_SEG) each, packed -- so that the backward launches a workgroup per slot in use instead of one per slot of
    // the (N >> CKPT_SHIFT) + T a sparse frame's layout has (the trained-scene profile: 98 000 workgroups that only found their
    // slot idle, ~12 us of its backward -- measured by doubling them).
    uint32_t dense_slots = 0xFFFFFFFFu;
    if (seg_first && !sparse && single_trip) {   // (dealt inside the trip above)
        if (threadIdx.x == 0) seg_first[num_tiles] = carry2;
        dense_slots = carry2;
    } else if (seg_first && !sparse) {   // (workgroup-uniform) more than 8 192 tiles: a second scan, over the ranges this workgroup wrote
        carry2 = 0;
        for (int base = 0; base < num_tiles; base += 1024 * SCAN_ITEMS) {
            const int t0 = base + threadIdx.x * SCAN_ITEMS;
            uint32_t need[SCAN_ITEMS], mine = 0;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) {
                const uint2 rg = t0 + k < num_tiles ? ranges[t0 + k] : make_uint2(0u, 0u);
                const uint32_t len = rg.y - rg.x;
                need[k] = len >= CKPT_DEEP_MIN ? (len + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT : 0u;
                mine += need[k];
            }
            const uint32_t inc = wave_inclusive_scan(mine);
            __syncthreads();   // (wsum's readers of the pass before are done)
            if (lane == 63) wsum[w] = inc;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = wsum[k];
                if (k < w) before += v;
                total += v;
            }
            uint32_t at = carry2 + before + inc - mine;
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) {
                if (t0 + k < num_tiles) seg_first[t0 + k] = at;
                at += need[k];
            }
            carry2 += total;
        }
        if (threadIdx.x == 0) seg_first[num_tiles] = carry2;
        dense_slots = carry2;
    }
    // The checkpoint buffer of a frame that was enqueued before its N was known is laid out for a GUESS of the slots it needs (round
    // 5: the shape's last count + a quarter -- not the (capacity >> CKPT_SHIFT) + T of the sparse layout, 128 bytes per list entry,
    // that a dense frame uses a tenth of).  A frame that needs more closes the gate exactly as one that overflows its binning
    // buffer does: the kernels behind return at once and the host runs it again, exactly sized.
    // (slots the frame's checkpoints need: the sparse layout's (N >> CKPT_SHIFT) + T, or a dense frame's packed count; 0: it leaves none)
    const uint32_t ckpt_needed = !seg_first || sparse_no_ckpt ? 0u : sparse ? (carry >> CKPT_SHIFT) + (uint32_t)num_tiles : dense_slots;
    if (seg_first && sa.ckpt_cap && threadIdx.x == 0 && ckpt_needed > sa.ckpt_cap) n_total[1] = 1u;
    uint32_t n_long = 0;
    if (any_long) {   // (workgroup-uniform) the frame has long lists: collect them
        for (int t00 = 0; t00 < num_tiles; t00 += 8 * 1024) {   // (eight loads in flight per thread: a 1080p frame is one trip)
            uint2 rg8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + u * 1024 + (int)threadIdx.x;
                rg8[u] = t < num_tiles ? ranges[t] : make_uint2(0u, 0u);   // (written by this workgroup before the barriers above)
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t00 + u * 1024 + (int)threadIdx.x;
                const bool is_long = rg8[u].y - rg8[u].x > threshold;
                const unsigned long long m = __ballot(is_long);
                if (m) {   // (wave-uniform; one atomic per wave)
                    uint32_t at = 0;
                    if (lane == (int)__builtin_ctzll(m)) at = atomicAdd(&n_long_sh, (uint32_t)__popcll(m));
                    at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)__builtin_ctzll(m));
                    if (is_long) large_tiles[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)t;
                }
            }
        }
        __syncthreads();
        n_long = n_long_sh;
    }
    if (threadIdx.x == 0) {
        n_total[2] = n_long;
        // (sparse << 63 | has-long-tiles << 62 | 30-bit ticket << 32 | N); words 1 and 2 of the slot: how many lists are long /
        // beyond SORT_CAP_MID entries -- the host sizes the next frame's launches by them
        const unsigned long long flags = ((unsigned long long)sparse << 31) | ((unsigned long long)any_long << 30);
        __hip_atomic_store(host_slot + 1, (unsigned long long)n_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_slot + 2, (unsigned long long)huge | ((unsigned long long)very_deep << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // word 3: checkpoint slots in use on a dense frame (0xFFFFFFFF: not a dense frame with checkpoints) -- the backward's grid
        // ... and, in the high half, the slots the frame's checkpoints need whatever its kind (the host sizes the next frame's buffer
        // by it and recognises a frame the checkpoint gate closed)
        __hip_atomic_store(host_slot + 3, (unsigned long long)dense_slots | ((unsigned long long)ckpt_needed << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // word 4: non-empty tiles, and (bit 32) whether the long tiles are blended split by depth -- launch-size hints for the shape's next frame
        // ... and (bit 33) a sparse frame that leaves no checkpoints: its backward runs without them
        __hip_atomic_store(host_slot + 4, (unsigned long long)n_nonempty | ((unsigned long long)(any_long ? deep_flag : 0u) << 32) |
                                              ((unsigned long long)(sparse_no_ckpt ? 1u : 0u) << 33), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_slot, ((flags | ticket) << 32) | carry, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void __launch_bounds__(1024) tile_scan_kernel(ScanArgs sa) { tile_scan_body<true>(sa); }

static ScanArgs make_scan_args(uint32_t* tile_count, int num_tiles, uint32_t* cell_count, int num_cells, uint2* ranges, uint32_t* cursor,
                               uint32_t* n_total, uint32_t* large_tiles, uint32_t* seg_first, uint32_t capacity,
                               unsigned long long* host_slot, uint32_t ticket, uint32_t ckpt_cap)
{
    // (both at most SORT_CAP_SMALL: that is what the one-workgroup-per-tile sort holds)
    const Switches& sw = switches();
    auto clamped = [](int v, int dflt) { v = v > 0 ? v : dflt; return (uint32_t)(v < 64 ? 64 : v > SORT_CAP_SMALL ? SORT_CAP_SMALL : v); };
    const uint32_t long_min_sparse = clamped(sw.long_min_sparse, LONG_MIN_SPARSE), long_min_dense = clamped(sw.long_min_dense, LONG_MIN_DENSE);
    const uint32_t dense_arg = long_min_dense | (sw.long_min_dense > 0 ? 0x80000000u : 0u);
    const uint32_t sparse_arg = long_min_sparse | (sw.long_min_sparse > 0 ? 0x80000000u : 0u);
    return ScanArgs{tile_count, num_tiles, cell_count, cell_count ? num_cells : 0, ranges, cursor, n_total, large_tiles, seg_first, capacity,
                    host_slot, ticket, sparse_arg, dense_arg, ckpt_cap, sw.frame_kind == 's' ? 1u : sw.frame_kind == 'd' ? 2u : 0u, (uint32_t)sw.deep_min};
}

void launch_tile_scan(uint32_t* tile_count, int num_tiles, uint32_t* cell_count, int num_cells, uint2* ranges, uint32_t* cursor,
                      uint32_t* n_total, uint32_t* large_tiles, uint32_t* seg_first, uint32_t capacity,
                      unsigned long long* host_slot, uint32_t ticket, uint32_t ckpt_cap, hipStream_t st)
{
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st,
                       make_scan_args(tile_count, num_tiles, cell_count, num_cells, ranges, cursor, n_total, large_tiles, seg_first, capacity, host_slot, ticket, ckpt_cap));
}

// ---------------------------------------------------------------------------------------------
// count / emit: bucket scatter plus a per-entry quad coverage mask.
//
// The value's top 4 bits carry a coverage mask: bit q is set when the splat can reach alpha >= 1/255 on
// some pixel of the tile's 8x8 quad q (q = qx + 2 qy).  It is CONSERVATIVE (may be set needlessly, never
// missing): the blend kernels skip a (quad, splat) pair whose bit is clear without touching a VGPR.
// max over the pixel-centre rectangle [x0,x0+7] x [y0,y0+7] of  f(d) = A dx^2 + B dx dy + C dy^2  (concave, d = centre - pixel).
// With the centre outside the rectangle the maximum sits on an edge FACING the centre: if the centre's x lies inside
// [x0, x0+7] the gradient condition rules the two vertical edges out (and likewise for y), so at most one vertical and one
// horizontal edge are candidates -- the nearer ones -- and each is a 1-D concave problem (vertex clamped to the segment).
// hA = -1/(2A), hC = -1/(2C): v_rcp_f32 (1 ulp) instead of an IEEE divide -- an error eps in the vertex position lowers the
// value found by ~A eps^2, far inside the threshold's slack.
__device__ __forceinline__ float max_power_in_quad(float sx, float sy, float A, float B, float C, float hA, float hC, float x0, float y0)
{
    const float dxl = sx - (x0 + 7.0f), dxh = sx - x0, dyl = sy - (y0 + 7.0f), dyh = sy - y0;  // d ranges over [dxl,dxh] x [dyl,dyh]
    const bool x_in = dxl <= 0.0f && dxh >= 0.0f, y_in = dyl <= 0.0f && dyh >= 0.0f;
    // nearer vertical edge: d.x = dxl when the centre is right of the quad (dxl > 0), else dxh (only used when !x_in)
    const float ex = dxl > 0.0f ? dxl : dxh;
    const float dy = fminf(dyh, fmaxf(dyl, B * ex * hC));
    const float fx = A * ex * ex + (B * ex + C * dy) * dy;
    const float ey = dyl > 0.0f ? dyl : dyh;
    const float dx = fminf(dxh, fmaxf(dxl, B * ey * hA));
    const float fy = C * ey * ey + (B * ey + A * dx) * dx;
    const float best = fmaxf(x_in ? -3.0e38f : fx, y_in ? -3.0e38f : fy);
    return (x_in && y_in) ? 0.0f : best;
}

// ---------------------------------------------------------------------------------------------
// Spatially coherent binning groups (the LDS binning path).  The preprocess kernel has counting-sorted the Gaussians by
// binning cell up to the last step: cell_count[c] = population of cell c, cell_slot[i] = (cell, slot inside the cell).
// cell_scatter: every workgroup prefix-sums the (<= BIN_MAX_CELLS) populations for itself and writes its Gaussians'
// indices to order[start(cell) + slot]; the total -- the Gaussians that touch a tile at all -- goes to windows[groups].x.
// Which Gaussian entry `tid` of binning group `block` is (P: none).  `tot` = (regular entries, where the big splats' groups begin
// -- the regular count rounded up to a whole group --, number of big splats, 0), written by cell_scatter_kernel: regular groups are
// runs of G entries of `order`; the first BIG_GROUPS_CAP groups behind them hold B big splats each in their first B slots; big
// splats beyond that fill whole groups.
__device__ __forceinline__ uint32_t big_groups_extent(const uint4 tot, uint32_t G, uint32_t B)
{
    const uint32_t n_big = tot.z;
    if (n_big == 0u || B == 0u) return tot.x;
    const uint32_t cap = (uint32_t)BIG_GROUPS_CAP * B;
    return n_big <= cap ? tot.y + (n_big + B - 1u) / B * G : tot.y + (uint32_t)BIG_GROUPS_CAP * G + (n_big - cap);
}
// Which binning group a workgroup of the count / emit launches takes: the big splats' groups FIRST (they span the whole screen and hold
// several times the pairs of a regular group: at the end of the grid they were the launch's tail), then the regular ones.
__device__ __forceinline__ uint32_t group_of_block(const uint4 tot, uint32_t G, uint32_t B, uint32_t block)
{
    if (tot.z == 0u || B == 0u) return block;
    const uint32_t big0 = tot.y / G, nb = (big_groups_extent(tot, G, B) + G - 1u) / G - big0;
    return block < nb ? big0 + block : block - nb < big0 ? block - nb : block;
}
__device__ __forceinline__ int group_member(const uint4 tot, uint32_t G, uint32_t B, uint32_t block, uint32_t tid, const uint32_t* __restrict__ order, int P)
{
    const uint32_t first = block * G;
    if (first < tot.y || tot.z == 0u || B == 0u) return first + tid < tot.x ? (int)order[first + tid] : P;
    const uint32_t bg = (first - tot.y) / G;
    if (bg < (uint32_t)BIG_GROUPS_CAP) {
        // (the B big splats of a padded group sit G / B slots apart: two or three per wave -- the count kernel walks a big splat's
        //  tiles with the WAVE that holds it, and B of them in the group's first wave were measured slower than round 4's spreading)
        const uint32_t stride = G >= B ? G / B : 1u, j = tid / stride;
        return (tid - j * stride == 0u && j < B && bg * B + j < tot.z) ? (int)order[first + tid] : P;
    }
    return (uint32_t)BIG_GROUPS_CAP * B + (first - tot.y - (uint32_t)BIG_GROUPS_CAP * G) + tid < tot.z ? (int)order[first + tid] : P;
}

__global__ void __launch_bounds__(256)
cell_scatter_kernel(int P, int num_cells, const uint32_t* __restrict__ cell_count, const uint2* __restrict__ cell_slot,
                    uint32_t* __restrict__ order, uint4* __restrict__ total_out, uint32_t G, uint32_t B)
{
    __shared__ uint32_t start[BIN_MAX_CELLS];
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int PER = BIN_MAX_CELLS / 256;  // consecutive cells per thread
    uint32_t c[PER], mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) c[k] = tid * PER + k < num_cells ? cell_count[tid * PER + k] : 0u, mine += c[k];
    const uint32_t n_big = cell_count[num_cells];   // the big cell (preprocess.hip)
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = wsum[k];
        if (k < w) before += v;
        total += v;
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < PER; ++k) start[tid * PER + k] = run, run += c[k];
    const uint32_t big_start = (total + G - 1u) / G * G;
    if (blockIdx.x == 0 && tid == 0) *total_out = make_uint4(total, big_start, n_big, 0u);
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i < P) {
        const uint2 cs = cell_slot[i];
        if (cs.x == (uint32_t)num_cells) {   // a big splat: slot s of the big cell
            const uint32_t s = cs.y, cap = (uint32_t)BIG_GROUPS_CAP * B;
            const uint32_t stride = G >= B ? G / B : 1u;
            order[s < cap ? big_start + (s / B) * G + (s % B) * stride : big_start + (uint32_t)BIG_GROUPS_CAP * G + (s - cap)] = (uint32_t)i;
        } else if (cs.x != 0xFFFFFFFFu)
            order[start[cs.x] + cs.y] = (uint32_t)i;
    }
}

// group_count: workgroup g owns entries [g G, (g + 1) G) of `order` -- Gaussians of one cell or of neighbouring cells.  It
// finds the tile window their rectangles span, counts the group's pairs per tile of the window in LDS, takes ONE returning
// atomic per touched tile on the global per-tile counters -- the value returned is where this group's run starts inside
// the tile's segment -- and leaves it in run_start[g][tile] for emit, which shares the partition.
template <bool H16>
__global__ void __launch_bounds__(BIN_GROUP)
group_count_kernel(int P, int G, Camera cam, const Splat* __restrict__ splats, const uint32_t* __restrict__ order,
                   uint4* __restrict__ windows, int groups, uint32_t* __restrict__ tile_count, uint32_t* __restrict__ run_start, uint32_t B)
{
    extern __shared__ uint32_t hist_words[];  // [num_tiles] counters (TileHist: words, or halves on frames beyond BIN_LDS_TILES tiles), only the window is used
    const TileHist<H16> hist{hist_words};
    __shared__ int win[4];              // min x, min y, max x (exclusive), max y (exclusive), in tiles
    const int NT = (int)blockDim.x, tid = threadIdx.x, lane = tid & 63;
    const uint4 tot = windows[groups];
    const uint32_t grp = group_of_block(tot, (uint32_t)G, B, blockIdx.x);
    const uint32_t first = grp * (uint32_t)G;
    if (first >= big_groups_extent(tot, (uint32_t)G, B)) {  // (uniform) nothing left for this group
        if (tid == 0) windows[grp] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    if (tid == 0) win[0] = win[1] = 0x7FFFFFFF, win[2] = win[3] = 0;
    SplatRect mine = load_rect(P, cam, splats, group_member(tot, (uint32_t)G, B, grp, (uint32_t)tid, order, P), false);
    // the window: wave-level min / max, then one LDS atomic per wave and bound
    int lo_x = mine.cnt ? mine.minx : 0x7FFFFFFF, lo_y = mine.cnt ? mine.miny : 0x7FFFFFFF;
    int hi_x = mine.cnt ? mine.minx + mine.width : 0, hi_y = mine.cnt ? mine.miny + (int)(mine.cnt / (uint32_t)mine.width) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo_x = min(lo_x, __shfl_xor(lo_x, d, 64)), lo_y = min(lo_y, __shfl_xor(lo_y, d, 64));
        hi_x = max(hi_x, __shfl_xor(hi_x, d, 64)), hi_y = max(hi_y, __shfl_xor(hi_y, d, 64));
    }
    __syncthreads();  // win initialised
    if (lane == 0 && hi_x > lo_x) atomicMin(&win[0], lo_x), atomicMin(&win[1], lo_y), atomicMax(&win[2], hi_x), atomicMax(&win[3], hi_y);
    __syncthreads();
    if (win[2] <= win[0]) {   // (uniform) a group without a Gaussian: the rounding between the regular and the big groups
        if (tid == 0) windows[grp] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const int wx0 = win[0], wy0 = win[1], ww = win[2] - win[0], wh = win[3] - win[1];
    const int n_win = ww * wh;  // (> 0: every Gaussian in `order` touches a tile)
    const float inv_ww = 1.0f / (float)ww;
    // tile of window index k: row = k / ww exactly (k < 2^22, see emit) without an integer divide
    auto tile_of = [&](int k) { const int r = (int)(((float)k + 0.5f) * inv_ww); return (wy0 + r) * cam.gx + wx0 + (k - r * ww); };
    for (int k = tid; k < n_win; k += NT) hist.zero(tile_of(k));
    // (round 6) a group of BIG splats (the padded groups behind the regular ones: at most B <= 64 members of 256 .. all tiles each) deals
    // every member's tiles over the WHOLE workgroup, as emit does: walked by the wave that holds it, 64 tiles a step, a 8 160-tile splat was
    // 128 steps of one wave while the other waves of the group had long finished theirs
    __shared__ int4 big_rect[64];
    __shared__ uint32_t n_big_members;
    const bool dealt = tot.z != 0u && B != 0u && first >= tot.y && (first - tot.y) / (uint32_t)G < (uint32_t)BIG_GROUPS_CAP && B <= 64u;   // (uniform)
    if (dealt && tid == 0) n_big_members = 0u;
    __syncthreads();
    if (!dealt)
        for_each_pair(mine, [&](int, int tx, int ty, const SplatRect&) { hist.add(ty * cam.gx + tx); });
    else {
        if (mine.cnt) big_rect[atomicAdd(&n_big_members, 1u)] = make_int4(mine.minx, mine.miny, mine.width, (int)mine.cnt);
        __syncthreads();
        const uint32_t members = n_big_members;
        for (uint32_t m = 0; m < members; ++m) {
            const int4 r = big_rect[m];
            const float inv_w = 1.0f / (float)r.z;
            for (int k = tid; k < r.w; k += NT) {
                const int ry = (int)(((float)k + 0.5f) * inv_w);   // k / width, exactly (k < 2^22)
                hist.add((r.y + ry) * cam.gx + r.x + (k - ry * r.z));
            }
        }
    }
    __syncthreads();
    // eight tiles per thread and round: the returning atomics of a round are all in flight together
    uint32_t* my_runs = run_start + (size_t)grp * (size_t)(cam.gx * cam.gy);
    for (int k0 = tid; k0 < n_win; k0 += 8 * NT) {
        uint32_t c[8], base[8];
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * NT;
            t[u] = k < n_win ? tile_of(k) : 0;
            c[u] = k < n_win ? hist.get(t[u]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) base[u] = c[u] ? atomicAdd(&tile_count[t[u]], c[u]) : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u]) my_runs[t[u]] = base[u];
    }
    if (tid == 0) windows[grp] = make_uint4((uint32_t)wx0, (uint32_t)wy0, (uint32_t)ww, (uint32_t)wh);
}

// (more than 64 KB of dynamic LDS has to be allowed once per kernel; the calls are idempotent and cheap)
static void allow_big_lds(const void* kernel, size_t dynamic_bytes)
{
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes) != hipSuccess) (void)hipGetLastError();   // (the launch reports what matters)
}

void launch_spatial_groups(int P, const Camera& cam, const Splat* splats, const uint32_t* cell_count, const uint2* cell_slot,
                           uint32_t* order, uint4* windows, uint32_t* tile_count, uint32_t* run_start, int group, int big_per_group, hipStream_t st)
{
    const int groups = (int)bin_groups_for(P, group);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, num_cells_of(cam.gx, cam.gy), cell_count,
                       cell_slot, order, windows + groups, (uint32_t)group, (uint32_t)big_per_group);
    const int num_tiles = cam.gx * cam.gy;
    if (num_tiles > BIN_LDS_TILES) {
        allow_big_lds((const void*)group_count_kernel<true>, TileHist<true>::bytes(num_tiles));
        hipLaunchKernelGGL(group_count_kernel<true>, dim3(groups), dim3(group), TileHist<true>::bytes(num_tiles), st, P, group, cam, splats,
                           order, windows, groups, tile_count, run_start, (uint32_t)big_per_group);
    } else {
        if (TileHist<false>::bytes(num_tiles) > 64 * 1024) allow_big_lds((const void*)group_count_kernel<false>, TileHist<false>::bytes(num_tiles));
        hipLaunchKernelGGL(group_count_kernel<false>, dim3(groups), dim3(group), TileHist<false>::bytes(num_tiles), st, P, group, cam, splats,
                           order, windows, groups, tile_count, run_start, (uint32_t)big_per_group);
    }
}

// count (fallback for frames with more than BIN_LDS_TILES tiles; otherwise group_count_kernel counts):
// tile_count[t] += number of Gaussians touching tile t, global atomics
__global__ void __launch_bounds__(BIN_THREADS)
count_kernel(int P, Camera cam, const Splat* __restrict__ splats, uint32_t* __restrict__ tile_count)
{
    const SplatRect mine = load_rect(P, cam, splats, blockIdx.x * BIN_GROUP + threadIdx.x, false);
    for_each_pair(mine, [&](int, int tx, int ty, const SplatRect&) { atomicAdd(&tile_count[ty * cam.gx + tx], 1u); });
}

// emit: every pair takes a slot of its tile's segment and stores its 64-bit sort key there.
// The value's low 4 bits carry a coverage mask: bit q is set when the splat can reach alpha >= 1/255 on some
// pixel of the tile's 8x8 quad q (q = qx + 2 qy).  It is CONSERVATIVE (may be set needlessly, never missing):
// the blend kernels skip a (quad, splat) pair whose bit is clear without touching a VGPR.
//
// A workgroup of 1024 threads owns one binning group (G <= 1024 Gaussians, the preprocess kernel's partition) and
// deals the group's pairs evenly to ALL its threads: the Gaussians' pair counts are prefix-summed, every Gaussian writes
// its index at the first of its pair slots in an LDS array, a forward max-fill turns that into "owner of slot s", and
// thread t then takes slots t, t + 1024, ... -- no per-pair search, no wave waiting for the one with the big splats, and
// for small inputs (the 6 890 Gaussians of the SMPL template are 108 groups of 64) sixteen waves per group instead of one.
// slot = segment start + this group's run start (both already known: tile_scan and the preprocess kernel's returning
// atomics) + a workgroup-private LDS counter -- one pass, no global atomic.  !USE_LDS: fallback, a global atomic per pair.
constexpr int EMIT_THREADS = 1024;
constexpr int EMIT_SLOTS = 8192;  // pair slots dealt per round (LDS: 2 bytes each)

// SCAN (round 5; frames of at most EMIT_SCAN_TILES tiles -- anything up to 1080p -- whose binning capacity is known before N is):
// there is no tile scan kernel in front -- every workgroup prefix-sums the frame's <= 32 KB of per-tile counts for itself (eight tiles
// per thread, one round trip that overlaps the group's record loads) where it used to read the scan's result, and decides the capacity gate
// from its own total; ONE extra workgroup of the launch (emit_scan_kernel) does what only one can do: ranges, N, flags, the long
// tiles' list, the checkpoint slot layout, the word the host polls.  The workgroup that arrives LAST at the launch's arrival
// counter re-zeroes the counters (the stand-alone scan kernel, their only reader, did that itself).  One launch and ~7 us of
// latency chain less on the frames that consist of nothing else (the human-only render).
constexpr int EMIT_SCAN_TILES = 8192;
constexpr int EMIT_SCAN_MAX_CHUNKS = 2;   // (round 6) ... frames of up to 16 384 tiles scan in two chunks of EMIT_SCAN_TILES (2048x1152, 2560x1440)

template <bool USE_LDS, bool SCAN, int SCAN_CHUNKS = 1, bool H16 = false>   // H16 (USE_LDS, !SCAN): 16-bit LDS counters, frames beyond BIN_LDS_TILES tiles
__device__ __forceinline__ void
emit_body(int P, int G, const Camera& cam, const Splat* __restrict__ splats, uint32_t* __restrict__ cursor,
          const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ order, const uint4* __restrict__ windows,
          int groups, uint64_t* __restrict__ keys, const uint32_t* __restrict__ gate, uint32_t* __restrict__ tile_count, uint32_t capacity,
          uint32_t* __restrict__ arrival, uint32_t big_per_group)
{
    if (!SCAN && *gate) return;  // binning buffer too small for this frame: the host re-runs it (hgs_api.hip)
    // BIN_BY_CELL (order != nullptr): the group is a run of `order` and spans a window of tiles; BIN_IN_ORDER: consecutive
    // Gaussians, the whole grid
    uint4 window = make_uint4(0u, 0u, (uint32_t)cam.gx, (uint32_t)cam.gy);
    uint4 tot = make_uint4(0u, 0u, 0u, 0u);
    uint32_t grp = blockIdx.x;   // the binning group this workgroup emits
    if (USE_LDS && order) {
        tot = windows[groups];
        grp = group_of_block(tot, (uint32_t)G, big_per_group, blockIdx.x);
        window = windows[grp];
        if (window.z == 0u) {  // the group is empty
            if (SCAN) {   // (it reads no counter, but the launch's arrival count includes it)
                __shared__ uint32_t last_empty;
                if (threadIdx.x == 0) last_empty = atomicAdd(arrival, 1u) == gridDim.x - 1u ? 1u : 0u;
                __syncthreads();
                if (last_empty) {
                    for (int t = threadIdx.x; t < cam.gx * cam.gy; t += EMIT_THREADS) tile_count[t] = 0u;
                    if (threadIdx.x == 0) atomicExch(arrival, 0u);
                }
            }
            return;
        }
    }
    constexpr int NT = EMIT_THREADS, PER = EMIT_SLOTS / NT;
    extern __shared__ uint32_t hist[];
    __shared__ float4 rec[BIN_GROUP][3];
    __shared__ uint32_t excl[BIN_GROUP];
    __shared__ __attribute__((aligned(16))) uint16_t own[EMIT_SLOTS];
    __shared__ uint32_t wtot[NT / 64];
    static_assert(sizeof(rec) + sizeof(excl) + sizeof(own) + sizeof(wtot) + sizeof(uint32_t) * BIN_LDS_TILES <= 160 * 1024 &&
                  sizeof(rec) + sizeof(excl) + sizeof(own) + sizeof(wtot) + sizeof(uint16_t) * BIN_LDS16_TILES + 64 <= 160 * 1024,
                  "emit's static LDS plus the per-tile array of the largest LDS-path frame must fit one CU's LDS");
    static_assert(!H16 || (USE_LDS && !SCAN), "the 16-bit counters belong to the plain LDS path");
    const TileHist<H16> hist16{hist};
    const int num_tiles = cam.gx * cam.gy;
    uint32_t* bins = USE_LDS ? hist : cursor;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g0 = (int)grp * G;
    SplatRect mine;
    mine.cnt = 0;
    // the group's Gaussians: entries of `order` (LDS path: neighbours on screen), else consecutive indices
    int gid = P;
    if (tid < G) gid = (USE_LDS && order) ? group_member(tot, (uint32_t)G, big_per_group, grp, (uint32_t)tid, order, P) : g0 + tid;
    if (tid < G) mine = load_rect(P, cam, splats, gid, true);  // (gid >= P: an empty rectangle)
    bool gated = false;          // SCAN: this frame needs more binning entries than it was given (workgroup-uniform, launch-uniform)
    uint32_t arrived_as = 0u;    // SCAN, thread 0: how many readers of the tile counters had reported in before this workgroup
    __shared__ uint32_t last_here;
    auto leave = [&]() {         // SCAN: the workgroup that reported in last re-zeroes the counters
        if (!SCAN) return;
        if (tid == 0) last_here = arrived_as == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (last_here) {
            for (int t = tid; t < num_tiles; t += NT) tile_count[t] = 0u;
            if (tid == 0) atomicExch(arrival, 0u);
        }
    };
    if (SCAN) {
        __shared__ uint32_t scan_w[NT / 64];
        __shared__ unsigned long long scan_total64;
        constexpr int SPT = EMIT_SCAN_TILES / NT;   // consecutive tiles per thread and chunk of the scan
        const uint32_t* my_runs = run_start + (size_t)grp * num_tiles;
        // where this group's runs begin inside the segments of its window's tiles (fetched now, added behind the scan: one round trip)
        const int ww = (int)window.z, n_win = ww * (int)window.w;
        const float inv_ww = 1.0f / (float)ww;
        auto tile_of = [&](int k) { const int r = (int)(((float)k + 0.5f) * inv_ww); return ((int)window.y + r) * cam.gx + (int)window.x + (k - r * ww); };
        uint32_t rw[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) rw[j] = tid + j * NT < n_win ? my_runs[tile_of(tid + j * NT)] : 0u;   // (meaningful only for the tiles this group touches)
        // (round 6) frames beyond EMIT_SCAN_TILES tiles -- 2048x1152: 9 216, 2560x1440: 14 400 -- scan in SCAN_CHUNKS chunks of that many, all
        // chunks' counts fetched up front (one round trip still), a running carry between them
        uint32_t c[SCAN_CHUNKS][SPT];
#pragma unroll
        for (int ch = 0; ch < SCAN_CHUNKS; ++ch) {
            const int t0 = ch * EMIT_SCAN_TILES + tid * SPT;   // (the counter array is 16-byte aligned and padded to a multiple of eight)
            uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = make_uint4(0u, 0u, 0u, 0u);
            if (t0 < num_tiles) lo = reinterpret_cast<const uint4*>(tile_count + t0)[0], hi = reinterpret_cast<const uint4*>(tile_count + t0)[1];
            c[ch][0] = lo.x, c[ch][1] = lo.y, c[ch][2] = lo.z, c[ch][3] = lo.w, c[ch][4] = hi.x, c[ch][5] = hi.y, c[ch][6] = hi.z, c[ch][7] = hi.w;
        }
        if (tid == 0) scan_total64 = 0ull;
        uint32_t carry = 0;
        unsigned long long m64 = 0ull;   // (the pair count again, in 64 bits: the 32-bit scan wraps silently beyond 2^32)
#pragma unroll
        for (int ch = 0; ch < SCAN_CHUNKS; ++ch) {
            const int t0 = ch * EMIT_SCAN_TILES + tid * SPT;
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < SPT; ++k) c[ch][k] = t0 + k < num_tiles ? c[ch][k] : 0u, sum += c[ch][k];
            const uint32_t inc = wave_inclusive_scan(sum);
            m64 += sum;
            __syncthreads();   // scan_total64 initialised / scan_w's readers of the chunk before are done
            if (lane == 63) scan_w[w] = inc;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) {
                const uint32_t v = scan_w[k];
                if (k < w) before += v;
                total += v;
            }
            uint32_t at = carry + before + inc - sum;
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                if (t0 + k < num_tiles) hist[t0 + k] = at;   // where every tile's segment begins
                at += c[ch][k];
            }
            carry += total;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m64 += (unsigned long long)__shfl_xor((long long)m64, d, 64);
        if (lane == 0 && m64) atomicAdd(&scan_total64, m64);
        // every reader of the counters reports in (its loads have returned: their values were used above); the last one re-zeroes
        // them -- at the END of its work: the returning atomic's round trip (~2 us) runs under the emission instead of in front of it
        if (tid == 0) arrived_as = atomicAdd(arrival, 1u);
        __syncthreads();
        gated = scan_total64 >= (unsigned long long)N_TOO_MANY || carry > capacity;
#pragma unroll
        for (int j = 0; j < SPT; ++j)
            if (tid + j * NT < n_win) hist[tile_of(tid + j * NT)] += rw[j];   // this group's cursor into its window's segments
        for (int k = tid + SPT * NT; k < n_win; k += NT) {   // (a window beyond EMIT_SCAN_TILES tiles: the big splats' groups of a large frame)
            const int t = tile_of(k);
            hist[t] += my_runs[t];
        }
    } else if (USE_LDS) {
        // this group's cursor into the tile segments of its window (entries of tiles the group does not touch are never
        // used, and run_start holds nothing meaningful for them)
        const uint32_t* my_runs = run_start + (size_t)grp * num_tiles;
        const int ww = (int)window.z, n_win = ww * (int)window.w;
        const float inv_ww = 1.0f / (float)ww;
        for (int k = tid; k < n_win; k += NT) {
            const int r = (int)(((float)k + 0.5f) * inv_ww);  // k / ww, exactly (k < 2^22)
            const int t = ((int)window.y + r) * cam.gx + (int)window.x + (k - r * ww);
            if (H16) hist16.zero(t);   // (the counter alone: segment start and run start are fetched per pair below)
            else hist[t] = cursor[t] + my_runs[t];
        }
    }
    // exclusive prefix sum of the pair counts over the group
    const uint32_t incl = wave_inclusive_scan(mine.cnt);
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) {
        const uint32_t v = wtot[k];
        if (k < w) before += v;
        total += v;
    }
    if (gated) {   // (every workgroup of the launch takes the same decision; the host re-runs the frame exactly sized)
        leave();
        return;
    }
    const uint32_t first = before + incl - mine.cnt;  // this Gaussian's pairs are slots [first, first + cnt)
    if (tid < G) {
        excl[tid] = first;
        rec[tid][0] = make_float4(mine.x, mine.y, mine.A, mine.B);
        rec[tid][1] = make_float4(mine.C, mine.thr, __uint_as_float(mine.depth_bits), 1.0f / (float)mine.width);
        rec[tid][2] = make_float4(__int_as_float(mine.minx), __int_as_float(mine.miny), __int_as_float(mine.width), __int_as_float(gid));
    }
    for (uint32_t base = 0; base < total; base += EMIT_SLOTS) {
        // owner of every slot of this round: scatter (index + 1) at each Gaussian's first slot, then fill forward (the
        // owners ascend with the slot, so "fill forward" is a running maximum)
        __syncthreads();  // previous round done with own[]
        reinterpret_cast<uint4*>(own)[tid] = make_uint4(0u, 0u, 0u, 0u);  // PER = 8 u16 per thread
        __syncthreads();
        if (mine.cnt && first < base + EMIT_SLOTS && first + mine.cnt > base) own[max(first, base) - base] = (uint16_t)(tid + 1);
        __syncthreads();
        {
            uint4 v = reinterpret_cast<uint4*>(own)[tid];
            uint32_t e[PER] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16, v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16};
#pragma unroll
            for (int k = 1; k < PER; ++k) e[k] = max(e[k], e[k - 1]);
            uint32_t run = e[PER - 1];  // inclusive max over the wave's threads
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t n = (uint32_t)__shfl_up((int)run, d, 64);
                if (lane >= d) run = max(run, n);
            }
            __syncthreads();  // wtot is free again
            if (lane == 63) wtot[w] = run;
            __syncthreads();
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < NT / 64; ++k)
                if (k < w) carry = max(carry, wtot[k]);
            const uint32_t up = (uint32_t)__shfl_up((int)run, 1, 64);
            carry = max(carry, lane ? up : 0u);  // everything before this thread's eight slots
#pragma unroll
            for (int k = 0; k < PER; ++k) e[k] = max(e[k], carry);
            reinterpret_cast<uint4*>(own)[tid] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
        __syncthreads();
        const uint32_t n_round = min((uint32_t)EMIT_SLOTS, total - base);
        for (uint32_t s = tid; s < n_round; s += NT) {
            const uint32_t o = (uint32_t)own[s] - 1u;
            const float4 a = rec[o][0], b = rec[o][1], c = rec[o][2];
            const uint32_t k = base + s - excl[o];
            // row of the pair inside the rectangle: k / width, exactly (k < 2^22) without an integer divide
            const uint32_t wdt = (uint32_t)__float_as_int(c.z);
            uint32_t ry = (uint32_t)(((float)k + 0.5f) * b.w);
            const uint32_t rx = k - ry * wdt;
            const int tx = __float_as_int(c.x) + (int)rx, ty = __float_as_int(c.y) + (int)ry;
            const float A = a.z, B = a.w, C = b.x, thr = b.y;
            uint32_t mask = 0xFu;
            if (A < 0.0f && C < 0.0f && 4.0f * A * C - B * B > 0.0f) {
                mask = 0;
                const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
                const float hA = -0.5f * __builtin_amdgcn_rcpf(A), hC = -0.5f * __builtin_amdgcn_rcpf(C);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (max_power_in_quad(a.x, a.y, A, B, C, hA, hC, x0 + (float)((q & 1) * 8), y0 + (float)((q >> 1) * 8)) >= thr)
                        mask |= 1u << q;
            }
            uint32_t slot;
            if (H16) {
                const int t = ty * cam.gx + tx;
                slot = cursor[t] + run_start[(size_t)grp * num_tiles + t] + hist16.add(t);
            } else
                slot = atomicAdd(&bins[ty * cam.gx + tx], 1u);
            // the entry IS its sort key: depth bits, then Gaussian index, with the mask riding in the low 4 bits
            keys[slot] = ((uint64_t)__float_as_uint(b.z) << 32) | (uint64_t)(((uint32_t)__float_as_int(c.w) << 4) | mask);
        }
    }
    leave();
}

template <bool USE_LDS, bool H16 = false>
__global__ void __launch_bounds__(EMIT_THREADS)
emit_kernel(int P, int G, Camera cam, const Splat* __restrict__ splats, uint32_t* __restrict__ cursor,
            const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ order, const uint4* __restrict__ windows,
            int groups, uint64_t* __restrict__ keys, const uint32_t* __restrict__ gate, uint32_t big_per_group)
{
    emit_body<USE_LDS, false, 1, H16>(P, G, cam, splats, cursor, run_start, order, windows, groups, keys, gate, nullptr, 0u, nullptr, big_per_group);
}

// grid = the binning groups + ONE workgroup (the last) that is the frame's tile scan
template <int SCAN_CHUNKS>
__global__ void __launch_bounds__(EMIT_THREADS)
emit_scan_kernel(int P, int G, Camera cam, const Splat* __restrict__ splats, const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ order,
                 const uint4* __restrict__ windows, int groups, uint64_t* __restrict__ keys, ScanArgs sa, uint32_t* __restrict__ arrival, uint32_t big_per_group)
{
    if ((int)blockIdx.x == groups) {
        tile_scan_body<false>(sa);
        __shared__ uint32_t last_here;
        __syncthreads();
        if (threadIdx.x == 0) last_here = atomicAdd(arrival, 1u) == gridDim.x - 1u ? 1u : 0u;
        __syncthreads();
        if (last_here) {
            for (int t = threadIdx.x; t < sa.num_tiles; t += EMIT_THREADS) sa.tile_count[t] = 0u;
            if (threadIdx.x == 0) atomicExch(arrival, 0u);
        }
        return;
    }
    emit_body<true, true, SCAN_CHUNKS>(P, G, cam, splats, nullptr, run_start, order, windows, groups, keys, nullptr, sa.tile_count, sa.capacity, arrival, big_per_group);
}

void launch_count(int P, const Camera& cam, const Splat* splats, uint32_t* tile_count, hipStream_t st)
{
    hipLaunchKernelGGL(count_kernel, dim3((P + BIN_GROUP - 1) / BIN_GROUP), dim3(BIN_THREADS), 0, st, P, cam, splats, tile_count);
}

void launch_emit(int P, const Camera& cam, const Splat* splats, uint32_t* cursor, const uint32_t* run_start, const uint32_t* order,
                 const uint4* windows, int group, int big_per_group, uint64_t* keys, const uint32_t* gate, hipStream_t st)
{
    if (group) {
        // (BIN_BY_CELL: the runs of `order` and the big splats' groups behind them; BIN_IN_ORDER: consecutive Gaussians)
        const int groups = order ? (int)bin_groups_for(P, group) : (P + group - 1) / group;
        const int num_tiles = cam.gx * cam.gy;
        if (num_tiles > BIN_LDS_TILES) {
            allow_big_lds((const void*)emit_kernel<true, true>, TileHist<true>::bytes(num_tiles));
            hipLaunchKernelGGL((emit_kernel<true, true>), dim3(groups), dim3(EMIT_THREADS), TileHist<true>::bytes(num_tiles), st,
                               P, group, cam, splats, cursor, run_start, order, windows, groups, keys, gate, (uint32_t)big_per_group);
        } else {
            if (TileHist<false>::bytes(num_tiles) > 64 * 1024) allow_big_lds((const void*)emit_kernel<true, false>, TileHist<false>::bytes(num_tiles));
            hipLaunchKernelGGL((emit_kernel<true, false>), dim3(groups), dim3(EMIT_THREADS), TileHist<false>::bytes(num_tiles), st,
                               P, group, cam, splats, cursor, run_start, order, windows, groups, keys, gate, (uint32_t)big_per_group);
        }
    } else
        hipLaunchKernelGGL(emit_kernel<false>, dim3((P + BIN_GROUP - 1) / BIN_GROUP), dim3(EMIT_THREADS), 0, st, P, BIN_GROUP, cam, splats,
                           cursor, nullptr, nullptr, nullptr, 0, keys, gate, 0u);
}

// (every emit workgroup repeats the scan: beyond ~1 000 of them -- a million Gaussians at 1080p, four rounds of workgroups -- the
//  repeats cost what the one-workgroup kernel does: 1 M Gaussians scan + emit 111.8 us apart, 112.4 folded; 500 k 74.6 / 71.3)
bool emit_scan_applies(int bin_mode, int num_tiles, int group, int P)
{
    if (!(bin_mode == BIN_IN_ORDER || bin_mode == BIN_BY_CELL) || group <= 0 || num_tiles > EMIT_SCAN_MAX_CHUNKS * EMIT_SCAN_TILES) return false;
    const size_t groups = bin_mode == BIN_BY_CELL ? bin_groups_for(P, group) : (size_t)((P + group - 1) / group);
    return groups <= 1024;
}

// order / windows: BIN_BY_CELL (the groups are runs of `order`, behind cell_scatter and group_count); nullptr: BIN_IN_ORDER
void launch_emit_scan(int P, const Camera& cam, const Splat* splats, const uint32_t* run_start, const uint32_t* order, const uint4* windows, int group,
                      int big_per_group, uint64_t* keys, uint32_t* tile_count, uint32_t* cell_count, int num_cells, uint2* ranges, uint32_t* cursor, uint32_t* n_total,
                      uint32_t* large_tiles, uint32_t* seg_first, uint32_t capacity, unsigned long long* host_slot, uint32_t ticket, uint32_t ckpt_cap,
                      uint32_t* arrival, hipStream_t st)
{
    const int groups = order ? (int)bin_groups_for(P, group) : (P + group - 1) / group, num_tiles = cam.gx * cam.gy;
    const ScanArgs sa = make_scan_args(tile_count, num_tiles, cell_count, num_cells, ranges, cursor, n_total, large_tiles, seg_first, capacity, host_slot, ticket, ckpt_cap);
    if (num_tiles <= EMIT_SCAN_TILES)
        hipLaunchKernelGGL(emit_scan_kernel<1>, dim3(groups + 1), dim3(EMIT_THREADS), sizeof(uint32_t) * num_tiles, st, P, group, cam, splats, run_start, order, windows,
                           groups, keys, sa, arrival, (uint32_t)big_per_group);
    else {
        static_assert(EMIT_SCAN_MAX_CHUNKS == 2, "one instance per chunk count");
        // (more than 64 KB of dynamic LDS has to be allowed once per kernel)
        allow_big_lds((const void*)emit_scan_kernel<2>, sizeof(uint32_t) * num_tiles);
        hipLaunchKernelGGL(emit_scan_kernel<2>, dim3(groups + 1), dim3(EMIT_THREADS), sizeof(uint32_t) * num_tiles, st, P, group, cam, splats, run_start, order, windows,
                           groups, keys, sa, arrival, (uint32_t)big_per_group);
    }
}

// ---------------------------------------------------------------------------------------------
// tile_sort: one workgroup per tile sorts the tile's segment and, while it still holds it, writes the tile's
// compacted per-quad lists the blend kernels stream (no bitmaps, no global prefix sum, no extra kernels).
// Sort key: (depth bits << 32) | (gaussian << 4) | mask: depth first, then Gaussian index (the mask rides along).
// Output: list[i] = (pos1 << 32) | (mask << 28 | gaussian), pos1 = 1-based position inside the tile.

__device__ __forceinline__ uint64_t list_entry(uint64_t key, uint32_t pos1)
{
    const uint32_t low = (uint32_t)key;
    return ((uint64_t)pos1 << 32) | (uint64_t)(((low & 15u) << GID_BITS) | (low >> 4));
}

// Small tiles (n <= 2048, i.e. practically all of them): the bitonic network runs in REGISTERS.  Thread t holds
// elements i = e * NT + t (e < E = m / NT; NT threads: 256 in the small-tile kernel, 1024 in the long-tile kernel).  A
// compare-exchange distance j >= NT pairs two registers of the same thread; 64 <= j < NT goes through LDS with
// barriers (3 of the 36 steps at m = NT = 256); j < 64 pairs two lanes of a
// wave: DPP moves for j = 1, 2 (quad_perm), 4 (row_shl:4 / row_shr:4 into complementary banks), 8 (row_ror:8),
// ds_bpermute with precomputed addresses for 16 and 32.
// The kernel is instruction-issue bound, so a step is kept to: fetch partner, ONE 64-bit compare, two selects.
// Which of the pair a lane keeps depends only on (lane bit j) == (element bit k), a compile-time lane pattern or a
// wave-uniform value: it lives in an SGPR pair, is combined with the compare's lane mask by s_xnor on the scalar
// unit, and drives VOP3 selects directly.
__device__ __forceinline__ uint32_t select32(unsigned long long m, uint32_t if_set, uint32_t if_clear)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp32(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, BANK_MASK, false);
}
// the key held by lane ^ J (J < 64); every lane of the wave must be active
template <uint32_t J>
__device__ __forceinline__ uint64_t partner_key(uint64_t key, uint32_t addr16, uint32_t addr32)
{
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t plo, phi;
    if constexpr (J == 1) plo = dpp32<0xB1>(lo, lo), phi = dpp32<0xB1>(hi, hi);           // quad_perm:[1,0,3,2]
    else if constexpr (J == 2) plo = dpp32<0x4E>(lo, lo), phi = dpp32<0x4E>(hi, hi);      // quad_perm:[2,3,0,1]
    else if constexpr (J == 4) {
        // banks 0,2 of each row read lane+4 (row_shl:4), banks 1,3 read lane-4 (row_shr:4)
        plo = dpp32<0x114, 0xA>(dpp32<0x104, 0x5>(lo, lo), lo);
        phi = dpp32<0x114, 0xA>(dpp32<0x104, 0x5>(hi, hi), hi);
    } else if constexpr (J == 8) plo = dpp32<0x128>(lo, lo), phi = dpp32<0x128>(hi, hi);  // row_ror:8
    else {
        const uint32_t addr = J == 16 ? addr16 : addr32;
        plo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)lo);
        phi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)hi);
    }
    return ((uint64_t)phi << 32) | plo;
}
// lanes whose index has bit J set
template <uint32_t J>
constexpr unsigned long long lane_bit_pattern()
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (l & J) m |= 1ull << l;
    return m;
}
// one compare-exchange of a lane's key with its partner's: keep_min_mask = lanes that keep the smaller key
__device__ __forceinline__ uint64_t keep_one(uint64_t a, uint64_t b, unsigned long long keep_min_mask)
{
    const unsigned long long lt = __builtin_amdgcn_uicmpl(a, b, 36);  // a < b (unsigned)
    const unsigned long long take_a = ~(lt ^ keep_min_mask);
    return ((uint64_t)select32(take_a, (uint32_t)(a >> 32), (uint32_t)(b >> 32)) << 32) |
           select32(take_a, (uint32_t)a, (uint32_t)b);
}

template <int E, uint32_t NT, uint32_t K, uint32_t J>
__device__ __forceinline__ void bitonic_step(uint64_t (&key)[E], uint64_t* sh, uint32_t addr16, uint32_t addr32)
{
    const uint32_t tid = threadIdx.x;
    if constexpr (J >= NT) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int f = e | (int)(J / NT);
            if (f != e && f < E) {
                const bool asc = (((uint32_t)e * NT + tid) & K) == 0u;
                const uint64_t a = key[e], b = key[f];
                if ((a > b) == asc) key[e] = b, key[f] = a;
            }
        }
    } else if constexpr (J >= 64u) {
        uint64_t other[E];
#pragma unroll
        for (int e = 0; e < E; ++e) sh[e * NT + tid] = key[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) other[e] = sh[e * NT + (tid ^ J)];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = (uint32_t)e * NT + tid;
            // bits J and K of i are both wave-uniform here (J, K >= 64)
            const bool keep_min = ((i & J) == 0u) == ((i & K) == 0u);
            key[e] = keep_one(key[e], other[e], __builtin_amdgcn_readfirstlane(keep_min) ? ~0ull : 0ull);
        }
    } else {
        constexpr unsigned long long PJ = lane_bit_pattern<J>();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            unsigned long long pk;  // lanes whose element index has bit K set
            if constexpr (K < 64u) pk = lane_bit_pattern<K>();
            else pk = __builtin_amdgcn_readfirstlane((((uint32_t)e * NT + tid) & K) != 0u) ? ~0ull : 0ull;
            const unsigned long long keep_min = ~(PJ ^ pk);  // bit J == bit K
            key[e] = keep_one(key[e], partner_key<J>(key[e], addr16, addr32), keep_min);
        }
    }
}

template <int E, uint32_t NT, uint32_t K, uint32_t J>
__device__ __forceinline__ void bitonic_merge(uint64_t (&key)[E], uint64_t* sh, uint32_t addr16, uint32_t addr32)
{
    bitonic_step<E, NT, K, J>(key, sh, addr16, addr32);
    if constexpr (J > 1u) bitonic_merge<E, NT, K, (J >> 1)>(key, sh, addr16, addr32);
}
template <int E, uint32_t NT, uint32_t K>
__device__ __forceinline__ void bitonic_phases(uint64_t (&key)[E], uint64_t* sh, uint32_t addr16, uint32_t addr32)
{
    if constexpr (K > 2u) bitonic_phases<E, NT, (K >> 1)>(key, sh, addr16, addr32);
    bitonic_merge<E, NT, K, (K >> 1)>(key, sh, addr16, addr32);
}

template <int E, uint32_t NT = 256u>
__device__ __forceinline__ void bitonic_in_registers(uint64_t (&key)[E], uint64_t* sh)
{
    const uint32_t lane = threadIdx.x & 63u;
    bitonic_phases<E, NT, NT * E>(key, sh, (lane ^ 16u) << 2, (lane ^ 32u) << 2);
}

// BUCKET SORT of one tile's keys by a workgroup of NT threads (E keys per thread, n <= E NT): the keys are dealt to E NT
// buckets by a monotone function of their depth bits (linear between the tile's nearest and farthest depth), an exclusive
// scan of the bucket sizes places the buckets, the keys are scattered into their buckets in LDS, and the key at every
// position then ranks itself among the handful of keys of its own bucket by the full 64-bit key (depth, then Gaussian
// index): a histogram pass, a scan, a scatter and a short ranking loop instead of a bitonic network of log^2 steps
// (4 096 keys on 1 024 threads: 31 us -> ~6 us; 512 keys on 256 threads: ~900 -> ~150 instructions per thread).
// In: key[e] = element e NT + tid (~0 beyond n).  Out: thread's e-th key `key[e]` is now the one that was scattered to LDS
// position e NT + tid and pos[e] its FINAL position in the sorted order (only for e NT + tid < n).  Returns false -- keys
// untouched -- when the depths pile up in one bucket (more than BUCKET_MAX keys in it, e.g. all depths equal): the caller
// takes the bitonic network.  sh_keys: E NT keys of LDS; bstart: E NT + 1 words; red: 3 NT / 64 words.
// EQ (round 4): EQUALISED buckets, for the lists whose depths cluster -- a person in front of a scene: nine keys in ten inside
// 3 % of the tile's depth range, where linear buckets overflow and the caller used to fall back to the bitonic network (70 us
// for a 4 000-entry list).  A coarse histogram (NT linear bins) is taken first; coarse bin b then owns as many fine buckets
// as it has keys, dealt linearly inside the bin: still monotone in the depth bits, n buckets in all.  The callers try the
// linear buckets first (no extra pass on the common path), then these, then the network.  coarse: 2 NT words of LDS.
constexpr uint32_t BUCKET_MAX = 48;
template <int E, uint32_t NT, bool EQ = false>
__device__ __forceinline__ bool bucket_sort(uint64_t (&key)[E], uint32_t (&pos)[E], uint32_t n, uint64_t* sh_keys, uint32_t* bstart,
                                            uint32_t* red, uint32_t* coarse = nullptr)
{
    constexpr uint32_t NB = (uint32_t)E * NT, NW = NT / 64u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * NT + tid;
        if (i < n) dmin = min(dmin, (uint32_t)(key[e] >> 32)), dmax = max(dmax, (uint32_t)(key[e] >> 32));
        bstart[i] = 0u;
    }
    if constexpr (EQ) coarse[tid] = 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, 64));
    }
    if (lane == 0) red[w] = dmin, red[NW + w] = dmax;
    if (tid == 0) bstart[NB] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < NW; ++k) dmin = min(dmin, red[k]), dmax = max(dmax, red[NW + k]);
    // bucket of a key: monotone in its depth bits (unsigned -> float conversion, a positive factor and the truncation are all
    // monotone), so bucket order never contradicts key order; what it does to ties is the ranking's business
    const float scale = (float)(EQ ? NT : NB) / ((float)(dmax - dmin) + 1.0f);
    if constexpr (EQ) {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((uint32_t)e * NT + tid < n) atomicAdd(&coarse[min((uint32_t)((float)((uint32_t)(key[e] >> 32) - dmin) * scale), NT - 1u)], 1u);
        __syncthreads();
        const uint32_t c = coarse[tid], inc = wave_inclusive_scan(c);
        if (lane == 63) red[2 * NW + w] = inc;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (uint32_t k = 0; k < NW; ++k)
            if (k < w) before += red[2 * NW + k];
        coarse[NT + tid] = before + inc - c;   // first fine bucket of coarse bin `tid`
        __syncthreads();
    }
    auto bucket_of = [&](uint64_t k) {
        const float f = (float)((uint32_t)(k >> 32) - dmin) * scale;
        if constexpr (EQ) {
            const uint32_t b = min((uint32_t)f, NT - 1u), c = coarse[b];
            return coarse[NT + b] + min((uint32_t)((f - (float)b) * (float)c), c - 1u);   // (c >= 1: the key itself is in bin b)
        } else
            return min((uint32_t)f, NB - 1u);
    };
    uint32_t slot[E], bkt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * NT + tid;
        bkt[e] = i < n ? bucket_of(key[e]) : 0u;
        slot[e] = i < n ? atomicAdd(&bstart[bkt[e]], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the NB bucket sizes (E consecutive buckets per thread) and the largest bucket
    uint32_t cnt[E], mine = 0, biggest = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) cnt[e] = bstart[tid * E + e], mine += cnt[e], biggest = max(biggest, cnt[e]);
    const uint32_t incl = wave_inclusive_scan(mine);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, d, 64));
    if (lane == 63) red[w] = incl;  // (every thread passed the barrier above after reading the min / max partials)
    if (lane == 0) red[2 * NW + w] = biggest;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t k = 0; k < NW; ++k) {
        if (k < w) before += red[k];
        biggest = max(biggest, red[2 * NW + k]);
    }
    if (biggest > BUCKET_MAX) {  // (workgroup-uniform)
        __syncthreads();         // nobody still reads `red` / `bstart` when the caller reuses the memory
        return false;
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int e = 0; e < E; ++e) bstart[tid * E + e] = run, run += cnt[e];
    if (tid == NT - 1u) bstart[NB] = run;  // == n
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((uint32_t)e * NT + tid < n) sh_keys[bstart[bkt[e]] + slot[e]] = key[e];
    __syncthreads();
    // position i holds some key of bucket b: its final position is the bucket's start plus its rank inside the bucket
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * NT + tid;
        pos[e] = i;
        if (i < n) {
            const uint64_t k = sh_keys[i];
            const uint32_t b = bucket_of(k), lo = bstart[b], hi = bstart[b + 1];
            uint32_t rank = 0;
            for (uint32_t j = lo; j < hi; ++j) rank += sh_keys[j] < k ? 1u : 0u;
            key[e] = k, pos[e] = lo + rank;
        }
    }
    return true;
}

// Appends up to 256 consecutive sorted entries of a tile (one per thread, `valid` when it exists) to the tile's
// NUM_LISTS compacted lists: list q < 4 keeps the entries covering quad q, list 4 those covering any quad.  List q
// of a tile lives at act[q * stride + s ...] (same offsets as the tile's segment of the sorted list, so no global
// prefix sum is needed).  carry[q] = entries already appended.  NW = waves in the workgroup (a chunk is 64 NW entries);
// sh32: >= NW * NUM_LISTS words of LDS.
template <int NW = 4>
__device__ __forceinline__ void compact_chunk(uint64_t entry, bool valid, uint32_t (&carry)[NUM_LISTS], uint32_t s,
                                              uint64_t* __restrict__ act, size_t stride, uint32_t* sh32)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t mask = valid ? ((uint32_t)entry >> GID_BITS) : 0u;
    uint32_t wrank[NUM_LISTS];
    bool flag[NUM_LISTS];
#pragma unroll
    for (int q = 0; q < NUM_LISTS; ++q) {
        flag[q] = q < 4 ? ((mask >> q) & 1u) != 0u : mask != 0u;
        const uint64_t m = __ballot(flag[q]);
        wrank[q] = (uint32_t)__popcll(m & lt);
        if (lane == 0) sh32[q * NW + w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if constexpr (NW <= 4) {
#pragma unroll
        for (int q = 0; q < NUM_LISTS; ++q) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)NW; ++k) {
                const uint32_t c = sh32[q * NW + k];
                if (k < w) before += c;
                total += c;
            }
            if (flag[q]) act[(size_t)q * stride + s + carry[q] + before + wrank[q]] = entry;
            carry[q] += total;
        }
    } else {
        // sixteen waves: lane k reads wave k's count and the wave scans them (5 LDS reads + shuffles, instead of 80 reads whose
        // results the scheduler keeps in 80 registers -- what pushed the long-tile kernels to 128 VGPRs and into scratch)
#pragma unroll
        for (int q = 0; q < NUM_LISTS; ++q) {
            const uint32_t c0 = lane < (uint32_t)NW ? sh32[q * NW + lane] : 0u;
            uint32_t c = c0;   // inclusive scan inside the row of 16 lanes: four DPP row shifts (zero shifted in), no LDS
            c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xf, 0xf, true);   // row_shr:1
            c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xf, 0xf, true);   // row_shr:2
            c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x114, 0xf, 0xf, true);   // row_shr:4
            if constexpr (NW > 8) c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x118, 0xf, 0xf, true);   // row_shr:8
            static_assert(NW == 16 || NW == 8, "one DPP row");
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)c, NW - 1);
            const uint32_t incl = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)w);    // (w is wave-uniform)
            const uint32_t own = (uint32_t)__builtin_amdgcn_readlane((int)c0, (int)w);
            if (flag[q]) act[(size_t)q * stride + s + carry[q] + (incl - own) + wrank[q]] = entry;
            carry[q] += total;
        }
    }
    __syncthreads();
}

// (returns the length of compacted list `want_list` -- the fused kernel's waves ask for their quad's)
template <int E>
__device__ __forceinline__ uint32_t tile_sort_small(uint32_t tile, uint32_t s, uint32_t n, const uint64_t* __restrict__ keys,
                                                    uint64_t* __restrict__ list, uint64_t* __restrict__ act, size_t stride,
                                                    uint32_t* __restrict__ act_count, uint64_t* sh, int want_list)
{
    uint64_t key[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 256u + threadIdx.x;
        key[e] = i < n ? keys[s + i] : ~0ull;
    }
    // 257 .. 1024 entries: the bucket sort (its keys and bucket array share the 16 KB the network's LDS steps use); the
    // bitonic network for up to 256 entries (one key per thread: 36 cheap steps), for 1025 .. 2048 (no room for the
    // buckets), and whenever the tile's depths pile up in one bucket
    bool sorted = false;  // (workgroup-uniform)
    if constexpr (E == 2 || E == 4) {
        uint32_t pos[E];
        uint32_t* bstart = reinterpret_cast<uint32_t*>(sh + E * 256);
        // (linear buckets; where the depths cluster, equalised ones; the bitonic network below when even those pile up)
        if (bucket_sort<E, 256u>(key, pos, n, sh, bstart, bstart + E * 256 + 4) ||
            bucket_sort<E, 256u, true>(key, pos, n, sh, bstart, bstart + E * 256 + 4, bstart + E * 256 + 4 + 16)) {
            __syncthreads();  // every thread has ranked its keys: the bucketed copy may be overwritten
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((uint32_t)e * 256u + threadIdx.x < n) sh[pos[e]] = key[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)e * 256u + threadIdx.x;
                key[e] = i < n ? sh[i] : ~0ull;
            }
            __syncthreads();  // (the compaction below reuses sh)
            sorted = true;
        }
    }
    if (!sorted) bitonic_in_registers<E>(key, sh);
    uint32_t carry[NUM_LISTS] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 256u + threadIdx.x;
        const uint64_t entry = list_entry(key[e], i + 1u);
        if (i < n) list[s + i] = entry;
        compact_chunk(entry, i < n, carry, s, act, stride, reinterpret_cast<uint32_t*>(sh));
    }
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < NUM_LISTS; ++q) {
        if (threadIdx.x == 0) act_count[tile * NUM_LISTS + q] = carry[q];
        if (q == want_list) mine = carry[q];
    }
    return mine;
}

// Tile sort FUSED with the forward blend: one workgroup per tile sorts the tile's segment, writes the sorted list and the
// five compacted lists, and its four waves then blend the tile's four quads from the lists they just wrote.  Besides
// saving a launch, this mixes the two phases on every CU -- the sort is latency-bound (loads, LDS exchanges, barriers:
// 64 % VALU-busy on its own), the blend VALU-bound (77 % on its own, waiting for scalar loads the rest): each fills the
// other's idle issue slots.  The lists are read through the scalar cache, which is not coherent with vector stores:
// the stores are released to L2 (agent-scope fence), the workgroup meets at a barrier, and the scalar cache is
// invalidated before the first list read (a neighbouring tile's prefetch may have pulled a stale line of this segment in).
// Long tiles (n > 2048) are left to tile_sort_large_kernel + the stand-alone forward blend over the long-tile list.
// FUSED = false: sort only (the stand-alone forward blend follows).
// Register budget of the fused kernel: 7 waves per SIMD, i.e. at most 72 VGPRs and 96 SGPRs -- the hardware adds 16 trap-handler
// SGPRs per wave, so 96 is where 7 waves end and 80 where 8 would begin (tools/microbench/occupancy_regs.hip; the compiler's own
// "Occupancy 8" for round 3's 95 SGPRs did not count them: that kernel ran 7 waves too).  Without the attribute the deep-worker
// path's scalar state raises the allocation to 106 SGPRs = 6 waves (C2: fused kernel +2.5 %); asking for 8 spills in the sort.
#ifndef FUSED_KERNEL_ATTR
#define FUSED_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(7)))
#endif
// (the kernel's arguments as one struct)
struct FusedKernelArgs {
    const uint2* ranges; const uint64_t* keys; uint64_t* list; uint64_t* act; size_t stride; uint32_t* act_count; const uint32_t* gate;
    Camera cam; uint32_t lastg; const Splat* splats; const float* bg; float* out_color; float* final_T; uint32_t* n_contrib;
    int clamp_output, long_sorted; Ckpt ck; const uint32_t* large_tiles; uint32_t num_workers;
};

// WORKERS = false: the instance for launches WITHOUT deep workers (no long-tile sort ran in front: the frames without long lists --
// the bench workload): the same tile path with the workers' code, LDS and registers compiled out.
template <bool FUSED, bool WORKERS = FUSED>
__global__ void __launch_bounds__(256) FUSED_KERNEL_ATTR
tile_sort_small_kernel(FusedKernelArgs a)
{
    __shared__ uint64_t sh[WORKERS ? DEEP_LDS_BYTES / 8 : (size_t)SORT_CAP_SMALL];  // (the deep workers' staging needs a little more than the sort)
    static_assert(DEEP_LDS_BYTES >= SORT_CAP_SMALL * 8, "sort buffer");
    const uint32_t* __restrict__ gate = a.gate;
    if (*gate) return;
    if (!WORKERS) a.num_workers = 0u;
    HGS_TRACE_PUT(0, wall_clock64());
    HGS_TRACE_PUT(3, ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));  // XCC_ID, HW_ID
    // (gate[7] = n_total[8], decided by the scan -- data, not launch sizes: 0 = no tile is blended split by depth, else the long tiles
    //  of MORE than that many entries are)
    const uint32_t deep_min = WORKERS && a.num_workers != 0u ? gate[7] : 0u;
    const bool deep_blend = deep_min != 0u;
    if (WORKERS && blockIdx.x < a.num_workers) {
        if (!deep_blend) return;
        // the first workgroups of the grid blend the long tiles -- sorted by the long tiles' kernels, which ran BEFORE this
        // kernel -- one (tile, quad) at a time, split by depth over their four waves (blend_fwd.h), beside the other tiles'
        // sort + blend
        deep_forward_worker(blockIdx.x, a.num_workers, a.cam, a.lastg, a.ranges, a.act, a.stride, a.act_count, a.splats, a.bg, a.out_color,
                            a.final_T, a.n_contrib, a.clamp_output, a.ck, a.large_tiles, gate - 1, *reinterpret_cast<DeepShared*>(sh), false);
        return;
    }
    const uint32_t tile_id = blockIdx.x - a.num_workers;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint2 rg = a.ranges[tile_id];
    const uint32_t s = rg.x, n = rg.y - rg.x;
    uint32_t* __restrict__ act_count = a.act_count;
    uint32_t mine = 0;
    if (n == 0) {
        if (threadIdx.x < NUM_LISTS) act_count[tile_id * NUM_LISTS + threadIdx.x] = 0u;
        if (!FUSED) return;
    } else if (n > gate[3]) {  // (gate points at n_total[1]: n_total[4] is this frame's long-tile threshold)
        // a long tile: sorted by the long tiles' kernels, which ran BEFORE this kernel (long_sorted) -- then only the blend is
        // left to do here -- or whose launch was skipped on the caller's guess that the frame has none: the tile's lists
        // read as empty and its pixels stay unwritten until the caller has repaired the guess (hgs_api.hip)
        if (!a.long_sorted) {
            if (threadIdx.x < NUM_LISTS) act_count[tile_id * NUM_LISTS + threadIdx.x] = 0u;
            return;
        }
        if (!FUSED) return;
        if (deep_blend && n > deep_min) {  // its quads are blended by the deep workers at the front of this grid; the tile marks its checkpoint slots
            ckpt_begin(a.ck, tile_id, n);
            return;
        }
        mine = ((const_u32p)act_count)[tile_id * NUM_LISTS + w];  // (HGS_DEEP_FORWARD=0: one wave per quad, as for any other tile)
    } else {
        if (n <= 256u) mine = tile_sort_small<1>(tile_id, s, n, a.keys, a.list, a.act, a.stride, act_count, sh, w);
        else if (n <= 512u) mine = tile_sort_small<2>(tile_id, s, n, a.keys, a.list, a.act, a.stride, act_count, sh, w);
        else if (n <= 1024u) mine = tile_sort_small<4>(tile_id, s, n, a.keys, a.list, a.act, a.stride, act_count, sh, w);
        else mine = tile_sort_small<8>(tile_id, s, n, a.keys, a.list, a.act, a.stride, act_count, sh, w);
        if (FUSED) {
            // The list stores must have been acknowledged by L2 (the vector L1 is write-through) before any wave reads
            // them through the scalar cache.  A workgroup-scope release fence does NOT wait for that on this target
            // (a workgroup shares its vector L1, so the compiler emits no vmcnt wait -- observed: one frame in a few
            // showed stale lists), and an AGENT-scope release also writes the XCD's whole L2 back (0.7 ms per frame when
            // every tile does it): spell the wait out.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_s_dcache_inv();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the invalidate has completed before the first list read
        }
    }
    HGS_TRACE_PUT(1, wall_clock64());
    HGS_TRACE_PUT(4, ((unsigned long long)n << 32) | mine);
    if (FUSED) {
        mine = __builtin_amdgcn_readfirstlane(mine);
        float4* ck_tile = ckpt_begin(a.ck, tile_id, n);
        const int lane = threadIdx.x & 63;
        const int tx = (int)(tile_id % (uint32_t)a.cam.gx), ty = (int)(tile_id / (uint32_t)a.cam.gx);
        const int px = tx * TILE + (w & 1) * 8 + (lane & 7), py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
        const bool inside = px < a.cam.W && py < a.cam.H;
        float4* ck_mine = ck_tile ? ck_tile + w * 64 + lane : nullptr;
        const uint64_t* my_list = a.act + (size_t)w * a.stride + s;
        // What the epilogue needs is put into VECTOR registers per lane before the walk -- output addresses, background, flags:
        // as scalar kernel arguments they sat in ~13 SGPRs across the blend loop, which this kernel (with the deep workers' state
        // in it) no longer has to spare (spill code inside the loop: +10 % on C4); fetched after the walk instead, each wave ended
        // with an exposed scalar-load round trip (+1.6 % on C2).  VGPRs are plentiful here: the sort's peak is elsewhere.
        const size_t HW = (size_t)a.cam.H * a.cam.W, pix = inside ? (size_t)py * a.cam.W + px : 0;
        float* o0 = a.out_color + pix;
        float* oT = a.final_T + pix;
        uint32_t* oN = a.n_contrib + pix;
        uint32_t* oP = a.ck.quad_nproc + tile_id * 4u + (uint32_t)w;
        float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
        uint32_t plane = (uint32_t)(HW * sizeof(float)), flags = (a.clamp_output ? 1u : 0u) | (inside ? 2u : 0u);
        asm volatile("" : "+v"(o0), "+v"(oT), "+v"(oN), "+v"(oP), "+v"(bg0), "+v"(bg1), "+v"(bg2), "+v"(plane), "+v"(flags));
        const FwdWalk r = blend_forward_walk(a.lastg, (float)px, (float)py, inside, mine, my_list, a.splats, ck_mine);
        if (ck_mine) {
            const uint32_t segs = (r.walked + (uint32_t)(CKPT_SEG - 1)) >> CKPT_SHIFT;
            if (segs >= 2u) ck_mine[(size_t)(segs - 1u) * 256u] = make_float4(r.T, r.C0, r.C1, r.C2);
            if (lane == 0) *oP = r.walked;
        }
        if (flags & 2u) {   // (the arithmetic of write_pixel, blend_fwd.h)
            const float Tf = __builtin_fabsf(r.T);
            *oT = Tf;
            float c0 = __builtin_fmaf(Tf, bg0, r.C0), c1 = __builtin_fmaf(Tf, bg1, r.C1), c2 = __builtin_fmaf(Tf, bg2, r.C2);
            uint32_t pass = 7u;
            if (flags & 1u) {
                pass = (c0 >= 0.0f && c0 <= 1.0f ? 1u : 0u) | (c1 >= 0.0f && c1 <= 1.0f ? 2u : 0u) | (c2 >= 0.0f && c2 <= 1.0f ? 4u : 0u);
                c0 = fminf(fmaxf(c0, 0.0f), 1.0f), c1 = fminf(fmaxf(c1, 0.0f), 1.0f), c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
            }
            *oN = r.last | (pass << 29);
            *o0 = c0;
            *reinterpret_cast<float*>(reinterpret_cast<char*>(o0) + plane) = c1;
            *reinterpret_cast<float*>(reinterpret_cast<char*>(o0) + 2u * (size_t)plane) = c2;
        }
    }
    HGS_TRACE_PUT(2, wall_clock64());
}

// ---- the long tiles' sorts (round 4) ----------------------------------------------------------------------------------
// A long tile (n > n_total[4]) is sorted BEFORE the fused kernel, whose deep workers blend it.  Three kernels share the work:
//   long_tile_plan_kernel   lists beyond SORT_CAP_MID entries are SPLIT BY DEPTH into parts of at most that many entries --
//                           several workgroups per long tile instead of one -- : one workgroup per such tile finds the depth
//                           range, counts the keys into PLAN_BUCKETS fine buckets (the linear, monotone bucket function of
//                           bucket_sort), assigns bucket b to part floor(exclusive_prefix(b) / PLAN_PART), counts every part's
//                           entries per quad list (so that a part knows where its compacted entries go) and appends one
//                           record per part to a device list.  Parts are contiguous in depth, so the concatenation of the
//                           sorted parts IS the sorted tile.  A tile whose depths pile up in a bucket (more than
//                           PLAN_BUCKET_MAX keys) is left to the one-workgroup fallback.
//   tile_sort_mid_kernel    one 512-thread workgroup per item: a long tile of at most SORT_CAP_MID entries, or one PART of a
//                           longer one (whose keys the plan kernel has moved into the part's stretch of `scratch`); bucket_sort,
//                           sorted list, compacted lists.  48 KB of LDS: two workgroups per CU.
//   tile_sort_large_kernel  the fallback for what the plan left (97 KB of LDS, one workgroup per CU, the CAP-sized chunk path
//                           for lists beyond LDS).
// f(i, key, valid) for every i < round_up(n, NT), NT threads, B independent loads in flight per thread: a plain `for (i = tid;
// i < n; i += NT)` around an LDS atomic or a ballot waits out one global-load latency per trip (measured: 20 us to plan a
// 3 400-entry tile with three such sweeps).  Every thread of the workgroup calls f the same number of times (ballots inside).
template <int B, int NT, typename F>
__device__ __forceinline__ void for_each_key(const uint64_t* __restrict__ k, uint32_t n, F f)
{
    for (uint32_t i0 = 0; i0 < n; i0 += (uint32_t)(B * NT)) {
        uint64_t v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t i = i0 + (uint32_t)u * NT + threadIdx.x;
            v[u] = i < n ? k[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t i = i0 + (uint32_t)u * NT + threadIdx.x;
            if (i0 + (uint32_t)u * NT < n) f(i, v[u], i < n);   // (uniform condition)
        }
    }
}
constexpr int PLAN_THREADS = 1024, PLAN_BUCKETS = 4096;
constexpr uint32_t PLAN_PART = 3072, PLAN_BUCKET_MAX = SORT_CAP_MID - PLAN_PART, PLAN_MAX_PARTS = 512;
constexpr uint32_t ACT_COUNT_FALLBACK = 0xFFFFFFFFu;   // act_count[tile][0] of a tile the plan left to the fallback kernel
struct SortPart {   // 64 bytes
    uint32_t tile, b_lo, b_hi, out_off, count, carry[NUM_LISTS], dmin;
    float scale;
    uint32_t pad[4];
};
static_assert(sizeof(SortPart) == 64, "SortPart layout");
__device__ __forceinline__ uint32_t plan_bucket_of(uint64_t key, uint32_t dmin, float scale)
{
    return min((uint32_t)((float)((uint32_t)(key >> 32) - dmin) * scale), (uint32_t)PLAN_BUCKETS - 1u);
}

__global__ void __launch_bounds__(PLAN_THREADS)
long_tile_plan_kernel(const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys, uint64_t* __restrict__ scratch,
                      uint32_t* __restrict__ act_count, const uint32_t* __restrict__ large_tiles, uint32_t* __restrict__ n_total,
                      SortPart* __restrict__ parts, uint32_t max_parts)
{
    __shared__ uint32_t fill[PLAN_MAX_PARTS];   // keys already moved to each part's stretch of `scratch`
    __shared__ uint32_t hist[PLAN_BUCKETS];          // bucket sizes, then their exclusive prefix
    __shared__ uint16_t part_of[PLAN_BUCKETS];
    __shared__ uint32_t part_b0[PLAN_MAX_PARTS + 1], part_off[PLAN_MAX_PARTS + 1];
    __shared__ uint32_t qcnt[PLAN_MAX_PARTS][NUM_LISTS];
    __shared__ uint32_t red[3 * (PLAN_THREADS / 64)];
    __shared__ uint32_t base_slot;
    if (n_total[1] || n_total[6] == 0u) return;  // gate; no list beyond SORT_CAP_MID entries on this frame (counted by the scan)
    const uint32_t threshold = max(n_total[4], (uint32_t)SORT_CAP_MID);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    constexpr uint32_t NW = PLAN_THREADS / 64, PER = PLAN_BUCKETS / PLAN_THREADS;
    for (uint32_t li = blockIdx.x; li < n_total[2]; li += gridDim.x) {
        const uint32_t tile = large_tiles[li];
        const uint2 rg = ranges[tile];
        const uint32_t s = rg.x, n = rg.y - rg.x;
        if (n <= threshold) continue;
        // depth range of the tile
        uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
        for_each_key<8, PLAN_THREADS>(keys + s, n, [&](uint32_t, uint64_t key, bool valid) {
            const uint32_t d = (uint32_t)(key >> 32);
            if (valid) dmin = min(dmin, d), dmax = max(dmax, d);
        });
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, d, 64));
            dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, d, 64));
        }
        if (lane == 0) red[w] = dmin, red[NW + w] = dmax;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) hist[tid * PER + k] = 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < NW; ++k) dmin = min(dmin, red[k]), dmax = max(dmax, red[NW + k]);
        const float scale = (float)PLAN_BUCKETS / ((float)(dmax - dmin) + 1.0f);
        for_each_key<8, PLAN_THREADS>(keys + s, n, [&](uint32_t, uint64_t key, bool valid) {
            if (valid) atomicAdd(&hist[plan_bucket_of(key, dmin, scale)], 1u);
        });
        __syncthreads();
        // exclusive prefix of the bucket sizes (PER consecutive buckets per thread), largest bucket
        uint32_t cnt[PER], mine = 0, biggest = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) cnt[k] = hist[tid * PER + k], mine += cnt[k], biggest = max(biggest, cnt[k]);
        const uint32_t incl = wave_inclusive_scan(mine);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, d, 64));
        __syncthreads();   // (everybody has read the min / max partials)
        if (lane == 63) red[w] = incl;
        if (lane == 0) red[2 * NW + w] = biggest;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (uint32_t k = 0; k < NW; ++k) {
            if (k < w) before += red[k];
            biggest = max(biggest, red[2 * NW + k]);
        }
        const uint32_t num_parts = (n - 1u) / PLAN_PART + 1u;   // (an upper bound: part ids are floor(prefix / PLAN_PART))
        if (biggest > PLAN_BUCKET_MAX || num_parts > PLAN_MAX_PARTS) {   // (workgroup-uniform) depths pile up: the fallback kernel
            if (tid == 0) act_count[tile * NUM_LISTS] = ACT_COUNT_FALLBACK, atomicAdd(&n_total[7], 1u);
            __syncthreads();
            continue;
        }
        for (uint32_t k = tid; k <= num_parts; k += PLAN_THREADS) part_b0[k] = 0xFFFFFFFFu, part_off[k] = n;
        for (uint32_t k = tid; k < num_parts; k += PLAN_THREADS) fill[k] = 0u;
        for (uint32_t k = tid; k < num_parts * NUM_LISTS; k += PLAN_THREADS) (&qcnt[0][0])[k] = 0u;
        __syncthreads();
        uint32_t run = before + incl - mine, prev_p = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t b = tid * PER + k, p = run / PLAN_PART;
            hist[b] = run;
            part_of[b] = (uint16_t)p;
            // first NON-EMPTY bucket of a part: where the part's entries start (a part id is only ever taken by a non-empty
            // bucket or shared with the next non-empty one, and prefixes grow by at most PLAN_BUCKET_MAX < PLAN_PART per bucket)
            // (only a bucket that can be its part's first non-empty one tries: the thread's first non-empty bucket, or one whose
            //  part differs from the thread's previous non-empty bucket's -- every lane trying cost 64-way conflicts)
            if (cnt[k] && p != prev_p) atomicMin(&part_b0[p], b), atomicMin(&part_off[p], run);
            if (cnt[k]) prev_p = p;
            run += cnt[k];
        }
        __syncthreads();
        // every part's entries per quad list, and the keys MOVED part by part into `scratch` (the part's stretch of the tile's
        // segment, any order inside it): the workgroup that sorts a part reads its keys contiguously instead of sweeping the
        // whole tile for them.  With a handful of parts every lane of a wave would hit the same few LDS counters (same-address
        // atomics of one instruction are serialised: 8 us for a 3 400-entry tile): the wave counts with ballots, part by part,
        // and adds once per (part, list); with many parts the direct atomics spread by themselves.
        for_each_key<8, PLAN_THREADS>(keys + s, n, [&](uint32_t, uint64_t key, bool valid) {
            const uint32_t p = valid ? part_of[plan_bucket_of(key, dmin, scale)] : 0xFFFFFFFFu, mask = valid ? (uint32_t)key & 15u : 0u;
            uint32_t dest = 0;
            if (num_parts <= 8u) {
                unsigned long long todo = __ballot(valid);
                while (todo) {   // (wave-uniform)
                    const int first = (int)__builtin_ctzll(todo);
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)p, first);
                    const bool mine_p = valid && p == p0;
                    const unsigned long long m = __ballot(mine_p);
                    uint32_t c = 0;
#pragma unroll
                    for (int q = 0; q < NUM_LISTS; ++q) {
                        const uint32_t cq = (uint32_t)__popcll(__ballot(mine_p && (q < 4 ? ((mask >> q) & 1u) != 0u : mask != 0u)));
                        if (lane == (uint32_t)q) c = cq;
                    }
                    if (lane < (uint32_t)NUM_LISTS && c) atomicAdd(&qcnt[p0][lane], c);
                    uint32_t base = 0;
                    if ((int)lane == first) base = atomicAdd(&fill[p0], (uint32_t)__popcll(m));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                    if (mine_p) dest = part_off[p0] + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    todo &= ~m;
                }
            } else if (valid) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((mask >> q) & 1u) atomicAdd(&qcnt[p][q], 1u);
                if (mask) atomicAdd(&qcnt[p][4], 1u);
                dest = part_off[p] + atomicAdd(&fill[p], 1u);
            }
            if (valid) scratch[s + dest] = key;
        });
        if (tid == 0) base_slot = atomicAdd(&n_total[5], num_parts);
        __syncthreads();
        // exclusive prefix over the parts, one wave per list (parts without entries -- ids no bucket took -- count as empty)
        if (w < (uint32_t)NUM_LISTS) {
            uint32_t carry = 0;
            for (uint32_t p0 = 0; p0 < num_parts; p0 += 64u) {
                const uint32_t p = p0 + lane, c = p < num_parts ? qcnt[p][w] : 0u;
                const uint32_t inc = wave_inclusive_scan(c);
                if (p < num_parts) qcnt[p][w] = carry + inc - c;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
            if (lane == 0) act_count[tile * NUM_LISTS + w] = carry;
        }
        __syncthreads();
        const uint32_t slot0 = base_slot;
        for (uint32_t p = tid; p < num_parts; p += PLAN_THREADS) {
            if (slot0 + p >= max_parts) continue;   // (cannot happen: the list is sized for N / PLAN_PART + tiles)
            SortPart r;
            r.tile = tile, r.dmin = dmin, r.scale = scale;
            r.out_off = min(part_off[p], n);
            // the part ends where the next part that HAS entries begins
            uint32_t nxt = p + 1u;
            while (nxt < num_parts && part_b0[nxt] == 0xFFFFFFFFu) ++nxt;
            const uint32_t end_off = nxt < num_parts ? part_off[nxt] : n;
            r.b_lo = part_b0[p] == 0xFFFFFFFFu ? 0u : part_b0[p];
            r.b_hi = part_b0[p] == 0xFFFFFFFFu ? 0u : (nxt < num_parts ? part_b0[nxt] : (uint32_t)PLAN_BUCKETS);
            r.count = part_b0[p] == 0xFFFFFFFFu ? 0u : end_off - r.out_off;
#pragma unroll
            for (int q = 0; q < NUM_LISTS; ++q) r.carry[q] = qcnt[p][q];
            r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0u;
            parts[slot0 + p] = r;
        }
        __syncthreads();   // LDS is reused by the next tile
    }
}

// Sorts ONE item with a workgroup of NT threads: `n` keys delivered by load_key(i), i < n (n <= CAP, or any n when the item is
// a whole tile and CAP > SORT_CAP_SMALL: the chunk path); the sorted entries go to list[s_dst + i] with positions pos_base + i + 1,
// the compacted entries behind carry[] in the tile's list slots (segment start s_tile).  Returns the final carry[] through
// `carry`.  sh: CAP keys, bucket_start: CAP + 1 words, red: 3 NT / 64 words.
#ifdef HGS_TRACE
#define MID_STAMP(slot) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (g_trace_buf && threadIdx.x == 0) g_trace_buf[(size_t)(6000u + blockIdx.x) * 8u + (slot)] = wall_clock64(); } while (0)
#else
#define MID_STAMP(slot) do { } while (0)
#endif
template <int CAP, int NT, typename LoadKey>
__device__ __forceinline__ void sort_item(LoadKey load_key, uint32_t n, uint32_t s_tile, uint32_t s_dst, uint32_t pos_base,
                                          uint32_t (&carry)[NUM_LISTS], uint64_t* __restrict__ list, uint64_t* __restrict__ act,
                                          size_t stride, uint64_t* sh, uint32_t* bucket_start, uint32_t* red, uint32_t* coarse)
{
    // The way out when the depths pile up in one bucket even after equalisation (all depths equal): the all-LDS bitonic network over the
    // keys' own LDS array.  Until round 5 this was the small tiles' register network with NT threads (E keys per thread, DPP below lane
    // distance 64): a fifth of the barriers, but its E = 8 instance is what held the long tiles' kernels at 128 VGPRs with 28-56 bytes of
    // scratch -- kernel arguments spilled at entry and reloaded on EVERY item's path -- for a path that real frames do not take.
    auto in_lds = [&]() {
        uint32_t m = 2;
        while (m < n) m <<= 1;   // (n <= CAP, a power of two)
        for (uint32_t i = threadIdx.x; i < m; i += NT) sh[i] = i < n ? load_key(i) : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t c = threadIdx.x; c < (m >> 1); c += NT) {
                    const uint32_t l = ((c & ~(j - 1u)) << 1) | (c & (j - 1u)), r = l | j;
                    const uint64_t a = sh[l], b = sh[r];
                    if ((a > b) == ((l & k) == 0u)) sh[l] = b, sh[r] = a;
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < n; i += NT) {
            const uint64_t entry = list_entry(sh[i], pos_base + i + 1u);
            list[s_dst + i] = entry;
            sh[i] = entry;   // (the compaction below reads the sorted list from here)
        }
    };
    // ---- bucket sort (the network above when the tile's depths pile up in one bucket), with as many
    // keys -- and buckets -- per thread as the list needs: a 1 100-entry list does not pay for 8 192 buckets ----
    auto by_buckets = [&](auto e_tag) {
        constexpr int E = decltype(e_tag)::value;
        uint64_t key[E];
        uint32_t pos[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = (uint32_t)e * NT + threadIdx.x;
            key[e] = i < n ? load_key(i) : ~0ull;
        }
        MID_STAMP(2);
        if (!bucket_sort<E, NT>(key, pos, n, sh, bucket_start, red) && !bucket_sort<E, NT, true>(key, pos, n, sh, bucket_start, red, coarse)) return false;
        MID_STAMP(3);
        __syncthreads();  // every thread has ranked its keys: the bucketed copy in LDS may be overwritten ...
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((uint32_t)e * NT + threadIdx.x < n) {
                const uint64_t entry = list_entry(key[e], pos_base + pos[e] + 1u);
                list[s_dst + pos[e]] = entry;
                sh[pos[e]] = entry;  // ... by the sorted list itself: the compaction below reads it from here
            }
        return true;
    };
    bool done;
    constexpr int EMAX = CAP / NT;
    if (n <= 1u * NT) done = by_buckets(std::integral_constant<int, 1>{});
    else if (EMAX >= 2 && n <= 2u * NT) done = by_buckets(std::integral_constant<int, (EMAX >= 2 ? 2 : 1)>{});
    else if (EMAX >= 4 && n <= 4u * NT) done = by_buckets(std::integral_constant<int, (EMAX >= 4 ? 4 : 1)>{});
    else done = by_buckets(std::integral_constant<int, EMAX>{});
    if (!done) in_lds();
    // the item is sorted, in LDS: compact it chunk by chunk (the bucket array is free by now: the compaction's scratch)
    __syncthreads();
    MID_STAMP(4);
    for (uint32_t base = 0; base < n; base += NT) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t entry = i < n ? sh[i] : 0ull;
        compact_chunk<NT / 64>(entry, i < n, carry, s_tile, act, stride, bucket_start);
    }
}

// A list LONGER than a workgroup's LDS, by ONE workgroup (the slow way out: the plan kernel normally splits such lists into parts
// for many workgroups): CAP-sized chunks are sorted in LDS into `scratch`, then every key finds its final position as its index
// in its own chunk plus, by binary search, the number of smaller keys in every other chunk (keys are distinct: they embed the
// Gaussian index).  O(n (n/CAP) log CAP) instead of a global merge network.  Then the compaction, from global memory.
template <int CAP, int NT>
__device__ __forceinline__ void sort_in_chunks(const uint64_t* __restrict__ src, uint32_t n, uint32_t s, uint64_t* __restrict__ list,
                                               uint64_t* __restrict__ scratch, uint64_t* __restrict__ act, size_t stride,
                                               uint32_t (&carry)[NUM_LISTS], uint64_t* sh, uint32_t* sh32)
{
    auto sort_in_lds = [&](const uint64_t* from, uint32_t count) {
        uint32_t m = 2;
        while (m < count) m <<= 1;
        for (uint32_t i = threadIdx.x; i < m; i += NT) sh[i] = i < count ? from[i] : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t c = threadIdx.x; c < (m >> 1); c += NT) {
                    const uint32_t l = ((c & ~(j - 1u)) << 1) | (c & (j - 1u)), r = l | j;
                    const uint64_t a = sh[l], b = sh[r];
                    if ((a > b) == ((l & k) == 0u)) sh[l] = b, sh[r] = a;
                }
                __syncthreads();
            }
    };
    const uint32_t chunks = (n + CAP - 1u) / CAP;
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint32_t c0 = c * CAP, cn = min((uint32_t)CAP, n - c0);
        sort_in_lds(src + c0, cn);
        for (uint32_t i = threadIdx.x; i < cn; i += NT) scratch[s + c0 + i] = sh[i];
        __syncthreads();
    }
    __threadfence_block();
    for (uint32_t i = threadIdx.x; i < n; i += NT) {
        const uint64_t ki = __builtin_nontemporal_load(&scratch[s + i]);
        const uint32_t own = i / CAP;
        uint32_t rank = i - own * CAP;
        for (uint32_t c = 0; c < chunks; ++c) {
            if (c == own) continue;
            const uint64_t* ch = scratch + s + c * CAP;
            uint32_t lo = 0, hi = min((uint32_t)CAP, n - c * CAP);  // first index whose key is >= ki
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (__builtin_nontemporal_load(&ch[mid]) < ki) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        list[s + rank] = list_entry(ki, rank + 1u);
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += NT) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t entry = i < n ? __builtin_nontemporal_load(&list[s + i]) : 0ull;
        compact_chunk<NT / 64>(entry, i < n, carry, s, act, stride, sh32);
    }
}

constexpr int SORT_MID_THREADS = 512;     // two or three such workgroups per CU whatever registers the body takes
constexpr int SORT_LARGE_THREADS = 1024;  // the fallback: a long tile is one workgroup's job, make it a big one
#define LARGE_SORT_ARGS const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys, uint64_t* __restrict__ list,            \
                        uint64_t* __restrict__ scratch, uint64_t* __restrict__ act, size_t stride, uint32_t* __restrict__ act_count, \
                        const uint32_t* __restrict__ large_tiles, const uint32_t* __restrict__ n_total, const SortPart* __restrict__ parts, \
                        int planned

__global__ void __launch_bounds__(SORT_MID_THREADS) __attribute__((amdgpu_waves_per_eu(4))) tile_sort_mid_kernel(LARGE_SORT_ARGS)   // (<= 128 VGPRs: two workgroups per CU)
{
    constexpr int CAP = SORT_CAP_MID, NT = SORT_MID_THREADS;
    __shared__ uint64_t sh[CAP];
    __shared__ uint32_t bucket_start[CAP + 1];  // bucket sizes, then (in place) their exclusive scan
    __shared__ uint32_t red[3 * (NT / 64)];
    __shared__ uint32_t coarse[2 * NT];
    MID_STAMP(0);
    if (n_total[1]) return;  // gate
    const uint32_t threshold = n_total[4], n_cand = n_total[2], n_items = n_cand + (planned ? n_total[5] : 0u);
    // a fixed grid walks the candidate tiles that tile_scan_kernel listed, then the parts the plan kernel made
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        uint32_t carry[NUM_LISTS] = {0, 0, 0, 0, 0};
        if (it < n_cand) {
            const uint32_t tile = large_tiles[it];
            const uint2 rg = ranges[tile];
            const uint32_t s = rg.x, n = rg.y - rg.x;
            if (n <= threshold || (planned && n > (uint32_t)CAP)) continue;   // (not long on this frame / the plan's or the fallback's)
            MID_STAMP(1);
#ifdef HGS_TRACE
            if (g_trace_buf && threadIdx.x == 0) g_trace_buf[(size_t)(6000u + blockIdx.x) * 8u + 7] = n;
#endif
            // (!planned: the plan and fallback kernels were not launched -- the stream's last frame had no list beyond CAP entries --
            //  and this frame has one after all: sorted here, slowly, this once)
            if (n > (uint32_t)CAP) sort_in_chunks<CAP, NT>(keys + s, n, s, list, scratch, act, stride, carry, sh, bucket_start);
            else sort_item<CAP, NT>([&](uint32_t i) { return keys[s + i]; }, n, s, s, 0u, carry, list, act, stride, sh, bucket_start, red, coarse);
#pragma unroll
            for (int q = 0; q < NUM_LISTS; ++q)
                if (threadIdx.x == 0) act_count[tile * NUM_LISTS + q] = carry[q];
            MID_STAMP(5);
        } else {
            const SortPart r = parts[it - n_cand];
            if (r.count == 0u) continue;
            const uint32_t s = ranges[r.tile].x;
            const uint32_t n = min(r.count, (uint32_t)CAP);
            const uint64_t* src = scratch + s + r.out_off;   // (the plan kernel moved the part's keys here)
#pragma unroll
            for (int q = 0; q < NUM_LISTS; ++q) carry[q] = r.carry[q];
            sort_item<CAP, NT>([&](uint32_t i) { return src[i]; }, n, s, s + r.out_off, r.out_off, carry, list, act, stride, sh, bucket_start, red, coarse);
        }
        __syncthreads();  // sh is reused by the next item
    }
}

// fallback: the tiles the plan kernel flagged (depths piling up in one of its buckets, or more parts than it can describe)
__global__ void __launch_bounds__(SORT_LARGE_THREADS) tile_sort_large_kernel(LARGE_SORT_ARGS)
{
    constexpr int CAP = SORT_CAP_LARGE, NT = SORT_LARGE_THREADS;
    __shared__ uint64_t sh[CAP];
    __shared__ uint32_t bucket_start[CAP + 1];
    __shared__ uint32_t red[3 * (NT / 64)];
    __shared__ uint32_t coarse[2 * NT];
    if (n_total[1] || n_total[7] == 0u) return;  // gate; the plan left nothing
    const uint32_t threshold = max(n_total[4], (uint32_t)SORT_CAP_MID);
    for (uint32_t li = blockIdx.x; li < n_total[2]; li += gridDim.x) {
        const uint32_t tile = large_tiles[li];
        const uint2 rg = ranges[tile];
        const uint32_t s = rg.x, n = rg.y - rg.x;
        if (n <= threshold || act_count[tile * NUM_LISTS] != ACT_COUNT_FALLBACK) continue;
        uint32_t carry[NUM_LISTS] = {0, 0, 0, 0, 0};
        if (n <= (uint32_t)CAP)
            sort_item<CAP, NT>([&](uint32_t i) { return keys[s + i]; }, n, s, s, 0u, carry, list, act, stride, sh, bucket_start, red, coarse);
        else
            sort_in_chunks<CAP, NT>(keys + s, n, s, list, scratch, act, stride, carry, sh, bucket_start);
#pragma unroll
        for (int q = 0; q < NUM_LISTS; ++q)
            if (threadIdx.x == 0) act_count[tile * NUM_LISTS + q] = carry[q];
        __syncthreads();  // sh is reused by the next tile
    }
}

void launch_tile_sort(const uint2* ranges, int num_tiles, const uint64_t* keys, uint64_t* list, uint64_t* scratch,
                      uint64_t* act, size_t stride, uint32_t* act_count, const uint32_t* large_tiles,
                      uint32_t* n_total, void* parts, uint32_t max_parts, bool small_tiles, bool long_tiles, const FusedBlend* fb,
                      const FrameHistory& hist, hipStream_t st)
{
    const bool plan_huge = hist.n_huge != 0;   // (-1: unknown)
    // The long-tile kernel goes FIRST: the small-tile kernel then blends (fused) the long tiles too, from the lists this
    // one wrote.  It walks a device-built list that is empty on most frames: the caller skips its launch when the previous
    // frame of this shape had no long tile, and runs it (and the long tiles' forward blend) later when that guess was wrong.
    if (long_tiles) {
        // Lists beyond SORT_CAP_MID entries are rare: the plan and fallback kernels are launched only when the stream's last frame
        // had one (or nothing is known); a frame that has one unannounced is sorted by the mid kernel's slow path, this once.
        const int wgs = num_tiles < 256 ? num_tiles : 256;
        if (plan_huge)
            hipLaunchKernelGGL(long_tile_plan_kernel, dim3(wgs), dim3(PLAN_THREADS), 0, st, ranges, keys, scratch, act_count, large_tiles, n_total,
                               (SortPart*)parts, max_parts);
        hipLaunchKernelGGL(tile_sort_mid_kernel, dim3(512), dim3(SORT_MID_THREADS), 0, st, ranges, keys, list, scratch, act, stride,
                           act_count, large_tiles, n_total, (const SortPart*)parts, plan_huge ? 1 : 0);
        if (plan_huge)
            hipLaunchKernelGGL(tile_sort_large_kernel, dim3(wgs), dim3(SORT_LARGE_THREADS), 0, st, ranges, keys, list, scratch, act, stride,
                               act_count, large_tiles, n_total, (const SortPart*)parts, 1);
    }
    if (small_tiles) {
        FusedKernelArgs ka{ranges, keys, list, act, stride, act_count, n_total + 1, Camera{}, 0u, nullptr, nullptr, nullptr, nullptr, nullptr,
                           0, long_tiles ? 1 : 0, Ckpt{}, large_tiles, 0u};
        if (fb) {
            // one worker workgroup per (long tile, quad) of the stream's last frame, and a quarter more: workgroups without an item
            // still occupy a slot while they find that out, in front of the tiles' workgroups
            uint32_t workers = long_tiles && deep_forward_enabled() ? deep_workers_for(num_tiles) : 0u;
            if (workers && hist.n_long >= 0) workers = min(workers, (uint32_t)(5 * hist.n_long + 64 + 7) & ~7u);
            // (the shape's last frame had long lists but blended them with one wave per quad -- the scan's many_flat_long: a handful of
            //  workers stand by; should this frame decide otherwise they walk all of its items, slowly and with the same result)
            if (workers && hist.deep_blend == 0) workers = 8u;
            ka.cam = fb->cam, ka.lastg = fb->lastg, ka.splats = fb->splats, ka.bg = fb->bg, ka.out_color = fb->out_color, ka.final_T = fb->final_T;
            ka.n_contrib = fb->n_contrib, ka.clamp_output = fb->clamp_output, ka.ck = fb->ck, ka.num_workers = workers;
            if (workers) hipLaunchKernelGGL((tile_sort_small_kernel<true, true>), dim3(workers + num_tiles), dim3(256), 0, st, ka);
            else hipLaunchKernelGGL((tile_sort_small_kernel<true, false>), dim3(num_tiles), dim3(256), 0, st, ka);
        } else
            hipLaunchKernelGGL(tile_sort_small_kernel<false>, dim3(num_tiles), dim3(256), 0, st, ka);
    }
}

}  // namespace hgs
