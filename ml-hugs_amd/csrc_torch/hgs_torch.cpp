// The PyTorch binding of the C ABI (include/hgs_rasterizer.h): what BASELINE.json's north_star calls "a thin C-ABI PyTorch
// extension".  Host-only code (g++ against the torch headers, __graft_entry__.build()); no kernels here -- PyTorch supplies
// device memory, streams and autograd, the library does the work.
//
// Round 5: ONE C++ call per frame.  A frame's host time (Python adapter + torch::autograd::Function plumbing + launches) was of the
// size of its kernels on the frames HUGS renders most (the human-only render: ~120-150 us of kernels), so the time of a step
// followed the host -- 0.156-0.213 ms box to box on the same kernels.  Three entry points now share one hand-made autograd node
// (a torch::autograd::Node, not a torch::autograd::Function: no IValue dictionary, no per-argument wrapping):
//   rasterize    the module API (GaussianRasterizer.forward, /root/reference/hugs/renderer/gs_renderer.py:144-152)
//   render       the whole of render() (gs_renderer.py:103-161): viewspace tensor, settings, rasterization, visibility -- one call
//   render_pair  the joint human+scene render AND the separate human-only render of one training step (gs_renderer.py:56,69) as
//                ONE node: the human-only frame runs on a library-side stream under the joint frame (forward and backward), its
//                gradients of the human tensors are added inside the joint frame's per-Gaussian kernel (hgs_backward_args.add_*),
//                and autograd sees one node on one stream -- no cross-stream AccumulateGrad fences, no elementwise sums.
// (A HIP graph of the frame's launches was measured first and dropped: tools/microbench/graph_launch.hip, DESIGN_HISTORY.md.)
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "hgs_rasterizer.h"

namespace {

using torch::Tensor;
using torch::autograd::SavedVariable;
using torch::autograd::variable_list;

// ------------------------------------------------------------------------------------------------ per-shape memory
struct Hint { int64_t n; bool has_long; bool sparse; int64_t ckpt_used = 0; uint64_t stamp = 0; };
using ShapeKey = std::tuple<int, int64_t, int64_t, int64_t>;   // (device, P, H, W)
uint64_t g_hint_clock = 0;
constexpr size_t HINT_SHAPES = 256;   // shapes remembered; the least recently used go first
std::mutex g_mu;
std::map<ShapeKey, Hint> g_hints;     // previous frame of each shape
// (caller holds g_mu)
void remember_hint(const ShapeKey& key, Hint h)
{
    h.stamp = ++g_hint_clock;
    g_hints[key] = h;
    while (g_hints.size() > HINT_SHAPES) {   // densification changes P all the time: drop the least recently used quarter
        std::vector<std::pair<uint64_t, ShapeKey>> by_age;
        for (auto& kv : g_hints) by_age.push_back({kv.second.stamp, kv.first});
        std::sort(by_age.begin(), by_age.end());
        for (size_t k = 0; k < HINT_SHAPES / 4; ++k) g_hints.erase(by_age[k].second);
    }
}
bool g_use_hint = true;
bool g_upstream_scale_grad = false;   // HGS_BWD_UPSTREAM_SCALE_GRAD on every backward (set_upstream_scale_grad)
bool g_use_ckpt = true;   // leave checkpoints for the depth-segmented backward on sparse frames (HGS_BWD_SEGMENTED=0: off)
thread_local int64_t t_last_n = -1, t_last_capacity = -1;
thread_local bool t_last_long = false, t_last_sparse = false;
thread_local int64_t t_last_ckpt_bytes = 0, t_last_ckpt_used = 0;

// + 12.5 % + 4096, rounded up to 1/16 of the next power of two (at least 64 Ki entries): the same number as
// diff_gaussian_rasterization._round_capacity -- frame after frame asks the caching allocator for the same size
int64_t round_capacity(int64_t n)
{
    const int64_t want = n + n / 8 + 4096;
    int64_t p2 = 1;
    while (p2 < want) p2 <<= 1;
    const int64_t granule = std::max<int64_t>(1 << 16, p2 >> 4);
    return (want + granule - 1) / granule * granule;
}

// checkpoint slots offered to a frame whose shape last used `used`: + 25 % + 64, in steps of 1 024 slots (4 MB) -- the same number as
// diff_gaussian_rasterization._round_ckpt_slots; a frame that needs more is detected on the device and run again
int64_t round_ckpt_slots(int64_t used)
{
    const int64_t want = used + used / 4 + 64;
    return (want + 1023) / 1024 * 1024;
}

// Scratch of frames that need no backward (the reference's validation / animation / canonical loops run under no_grad,
// gs_trainer.py:448-684): one persistent arena per (device, stream) instead of an allocation per frame -- work on a stream
// is ordered, so the next frame on that stream may overwrite it.  (The Python binding keeps the same policy: _arena.)
// Keyed by the host thread too: two threads issuing frames on one stream interleave their enqueues, and the second frame's
// first kernel would overwrite scratch the first frame's later kernels have yet to read.
struct Arena { Tensor t; uint64_t stamp = 0; };
std::map<std::tuple<int, void*, std::thread::id>, Arena> g_arenas;
uint64_t g_arena_clock = 0;
constexpr size_t MAX_ARENAS = 32;   // threads and streams come and go: the least recently used arenas are let go
Tensor arena_for(int dev, void* stream, size_t bytes, const at::TensorOptions& bopts)
{
    std::lock_guard<std::mutex> lk(g_mu);
    const auto key = std::make_tuple(dev, stream, std::this_thread::get_id());
    Arena& a = g_arenas[key];
    a.stamp = ++g_arena_clock;
    while (g_arenas.size() > MAX_ARENAS) {
        auto oldest = g_arenas.begin();
        for (auto it = g_arenas.begin(); it != g_arenas.end(); ++it)
            if (it->second.stamp < oldest->second.stamp) oldest = it;
        g_arenas.erase(oldest);   // (never `a`: it carries the newest stamp)
    }
    Tensor& t = g_arenas[key].t;
    if (!t.defined() || (size_t)t.numel() < bytes) t = at::empty({(int64_t)(bytes + bytes / 4 + 4096)}, bopts);
    return t;
}

// The gradient sink of render() -- `viewspace_points`, a zero [P,3] leaf that requires grad (gs_renderer.py:107-113) -- without a
// fill kernel per frame: every frame's tensor is a fresh leaf over the SAME zero-filled storage (one per device, grown on demand).
// Nothing writes into it: the rasterizer never reads means2D, and autograd refuses in-place operations on a leaf that requires grad.
std::map<std::pair<int, int>, Tensor> g_zeros;   // per (device, dtype): the reference's zeros_like(means3D) follows means3D's dtype
Tensor viewspace_zeros(int64_t rows, const at::TensorOptions& fopts)
{
    Tensor base;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Tensor& z = g_zeros[{(int)fopts.device().index(), (int)c10::typeMetaToScalarType(fopts.dtype())}];
        if (!z.defined() || z.size(0) < rows) z = at::zeros({std::max<int64_t>(rows + rows / 4 + 1024, 4096), 3}, fopts);
        base = z;
    }
    Tensor v = base.narrow(0, 0, rows).detach();
    v.set_requires_grad(true);
    return v;
}

// the library-side stream the second frame of a pair runs on, and two events per host thread to fence it against the caller's
std::map<int, c10::hip::HIPStream> g_side;
c10::hip::HIPStream side_stream(int dev)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_side.find(dev);
    if (it == g_side.end()) it = g_side.emplace(dev, c10::hip::getStreamFromPool(/*isHighPriority=*/false, (c10::DeviceIndex)dev)).first;
    return it->second;
}
struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    void make()
    {
        if (a) return;
        TORCH_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess,
                    "hipEventCreate failed");
    }
};
thread_local std::map<int, Events> t_events;
Events& events_for(int dev)
{
    Events& e = t_events[dev];
    e.make();
    return e;
}
// `later` does not start before what `earlier` holds now has run
void fence(hipEvent_t ev, hipStream_t earlier, hipStream_t later)
{
    TORCH_CHECK(hipEventRecord(ev, earlier) == hipSuccess && hipStreamWaitEvent(later, ev, 0) == hipSuccess, "stream fence failed");
}

inline const float* fptr(const Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

inline Tensor f32c(const Tensor& t)
{
    if (!t.defined() || t.numel() == 0) return Tensor();
    if (t.scalar_type() == at::kFloat && t.is_contiguous()) return t;
    return t.to(at::kFloat).contiguous();
}

inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

struct AllocCtx { std::vector<Tensor>* keep; at::TensorOptions opts; };
void* alloc_cb(void* ctx, int, size_t bytes)
{
    auto* a = static_cast<AllocCtx*>(ctx);
    a->keep->push_back(at::empty({(int64_t)bytes}, a->opts));
    return a->keep->back().data_ptr();
}

void raise_last(const char* what) { TORCH_CHECK(false, what, ": ", hgs_last_error()); }

// element offsets inside the gradient slab: accumulator [P + P2,12], means2D [P + P2,3]; opacity, colors, means3D, cov3D, sh,
// scales, rotations of the first set; then of the second set (hgs_segment) in the same order
struct GradLayout {
    int64_t off[16], size[16], total;
    GradLayout() : total(0) {}
    GradLayout(int64_t P, int64_t M, int64_t P2, int64_t M2)
    {
        const int64_t Pt = P + P2;
        const int64_t s[16] = {12 * Pt, 3 * Pt, P, 3 * P, 3 * P, 6 * P, 3 * M * P, 3 * P, 4 * P,
                               P2, 3 * P2, 3 * P2, 6 * P2, 3 * M2 * P2, 3 * P2, 4 * P2};
        total = 0;
        for (int k = 0; k < 16; ++k) size[k] = s[k], off[k] = total, total += (s[k] + 63) / 64 * 64;
        if (total < 1) total = 1;
    }
};

void point_at_grads(hgs_backward_args& bw, float* base, const GradLayout& g, int64_t M, int64_t M2)
{
    bw.grad_accum = base + g.off[0], bw.dL_dmeans2D = base + g.off[1], bw.dL_dopacity = base + g.off[2];
    bw.dL_dcolors = base + g.off[3], bw.dL_dmeans3D = base + g.off[4], bw.dL_dcov3D = base + g.off[5];
    bw.dL_dsh = M ? base + g.off[6] : nullptr;
    bw.dL_dscales = base + g.off[7], bw.dL_drotations = base + g.off[8];
    bw.seg2_dL_dopacity = base + g.off[9], bw.seg2_dL_dcolors = base + g.off[10], bw.seg2_dL_dmeans3D = base + g.off[11];
    bw.seg2_dL_dcov3D = base + g.off[12], bw.seg2_dL_dsh = M2 ? base + g.off[13] : nullptr;
    bw.seg2_dL_dscales = base + g.off[14], bw.seg2_dL_drotations = base + g.off[15];
}

// ------------------------------------------------------------------------------------------------ one render's inputs
// One set of Gaussians as the library takes it (fp32, contiguous; undefined = absent)
struct Set {
    Tensor means3D, sh, colors, opac, scales, rot, cov;
    int64_t P() const { return means3D.defined() ? means3D.size(0) : 0; }
    int64_t M() const { return sh.defined() && sh.numel() ? sh.size(1) : 0; }
};

Set make_set(const Tensor& means3D, const Tensor& sh, const Tensor& colors, const Tensor& opac, const Tensor& scales, const Tensor& rot,
             const Tensor& cov)
{
    return Set{f32c(means3D), f32c(sh), f32c(colors), f32c(opac), f32c(scales), f32c(rot), f32c(cov)};
}

// (the second set goes to the kernels as raw pointers: a tensor on another device, of another dtype or with another row count
//  than means3D would be read -- or its gradient written -- out of bounds instead of raising)
void check_segment_tensor(const char* name, const Tensor& t, const Tensor& means3D, int64_t tail, bool three_d = false)
{
    if (!t.defined() || t.numel() == 0) return;
    TORCH_CHECK(t.device() == means3D.device(), "second: ", name, " is on ", t.device(), ", means3D on ", means3D.device());
    TORCH_CHECK(t.scalar_type() == at::kFloat, "second: ", name, " must be float32");
    TORCH_CHECK(t.size(0) == means3D.size(0), "second: ", name, " has ", t.size(0), " rows, means3D has ", means3D.size(0));
    if (three_d) {
        TORCH_CHECK(t.dim() == 3 && t.size(2) == 3, "second: ", name, " must have dimensions (num_points, M, 3)");
    } else if (tail > 0) {
        TORCH_CHECK(t.numel() == means3D.size(0) * tail, "second: ", name, " must have dimensions (num_points, ", tail, ")");
    }
}

void fill_segment(hgs_segment& g, const Set& s)
{
    memset(&g, 0, sizeof g);
    if (s.P() == 0) return;
    TORCH_CHECK(s.means3D.dim() == 2 && s.means3D.size(1) == 3, "second: means3D must have dimensions (num_points, 3)");
    check_segment_tensor("shs", s.sh, s.means3D, 0, true);
    check_segment_tensor("colors_precomp", s.colors, s.means3D, 3);
    check_segment_tensor("opacities", s.opac, s.means3D, 1);
    check_segment_tensor("scales", s.scales, s.means3D, 3);
    check_segment_tensor("rotations", s.rot, s.means3D, 4);
    check_segment_tensor("cov3D_precomp", s.cov, s.means3D, 6);
    g.P = (int32_t)s.P();
    g.M = (int32_t)s.M();
    g.means3D = fptr(s.means3D), g.shs = fptr(s.sh), g.colors_precomp = fptr(s.colors), g.opacities = fptr(s.opac);
    g.scales = fptr(s.scales), g.rotations = fptr(s.rot), g.cov3D_precomp = fptr(s.cov);
}

struct Settings {
    Tensor bg, view, proj, campos;   // on the device, fp32, contiguous
    int64_t H, W, degree;
    double tanfovx, tanfovy, mod;
    bool prefiltered, debug, clamp_output, with_visibility;
};

// One hgs_rasterize_forward / hgs_rasterize_backward pair and everything its backward needs
struct Frame {
    hgs_backward_args bw;     // bw.fwd and bw.state are filled by the forward; the gradient pointers aim into `slab`
    Set a, b;                 // the first and (optional) second set of Gaussians
    Settings s;
    Tensor color, radii, visible;
    Tensor scratch, slab;
    std::vector<Tensor> keep;   // buffers the allocation callback handed out (an unhinted or under-guessed frame)
    GradLayout gl;
    ShapeKey key;
    bool fresh = true;        // the slab's accumulator was zeroed by this frame's forward and has not been used yet
    bool deferred = false;
    Frame() { memset(&bw, 0, sizeof bw); }
    int64_t P1() const { return a.P(); }
    int64_t P2() const { return b.P(); }
};

void fill_forward(Frame& f)
{
    hgs_forward_args& a = f.bw.fwd;
    const Settings& s = f.s;
    a.s.image_height = (int32_t)s.H, a.s.image_width = (int32_t)s.W;
    a.s.tanfovx = (float)s.tanfovx, a.s.tanfovy = (float)s.tanfovy;
    a.s.bg = fptr(s.bg), a.s.viewmatrix = fptr(s.view), a.s.projmatrix = fptr(s.proj), a.s.campos = fptr(s.campos);
    a.s.scale_modifier = (float)s.mod, a.s.sh_degree = (int32_t)s.degree, a.s.prefiltered = s.prefiltered, a.s.debug = s.debug;
    a.P = (int32_t)f.a.P();
    a.M = (int32_t)f.a.M();
    a.means3D = fptr(f.a.means3D), a.shs = fptr(f.a.sh), a.colors_precomp = fptr(f.a.colors), a.opacities = fptr(f.a.opac);
    a.scales = fptr(f.a.scales), a.rotations = fptr(f.a.rot), a.cov3D_precomp = fptr(f.a.cov);
    a.clamp_output = s.clamp_output ? 1 : 0;
    fill_segment(a.seg2, f.b);
}

// hgs_forward_args.before_wait: the node is attached to the graph (edges, saved inputs, the image's grad_fn) while the GPU works towards
// N, not after N has arrived -- from there to the backward's launch the GPU has ~45 us of forward left on a human-only frame
struct BeforeWait { std::function<void()> fn; bool done = false; };
void before_wait_cb(void* ctx)
{
    auto* b = static_cast<BeforeWait*>(ctx);
    if (b->fn && !b->done) b->done = true, b->fn();
}

// Allocate the frame's outputs and scratch, enqueue its forward on `stream`.  `defer`: do not wait for N when the shape has a
// history (finish_frame() must follow).  Returns with f.bw.state filled (num_rendered = -1 for a deferred frame).
void start_frame(Frame& f, bool needs_grad, hipStream_t stream, bool defer, BeforeWait* before_wait = nullptr)
{
    const auto dev = f.a.means3D.device();
    const int64_t P1 = f.P1(), P2 = f.P2(), P = P1 + P2, H = f.s.H, W = f.s.W;
    TORCH_CHECK(P2 == 0 || P1 > 0, "a second set of Gaussians needs a non-empty first one");
    TORCH_CHECK(P2 == 0 || f.b.means3D.device() == dev, "second: means3D is on ", f.b.means3D.device(), ", the first set of Gaussians on ", dev);
    TORCH_CHECK(P2 == 0 || (f.b.means3D.dim() == 2 && f.b.means3D.size(1) == 3), "means3D must have dimensions (num_points, 3)");
    const auto fopts = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto bopts = at::TensorOptions().dtype(at::kByte).device(dev);
    f.color = P == 0 ? at::zeros({3, H, W}, fopts) : at::empty({3, H, W}, fopts);
    f.radii = at::empty({P}, fopts.dtype(at::kInt));
    if (f.s.with_visibility) f.visible = at::empty({P}, fopts.dtype(at::kBool));   // `radii > 0`, written with radii
    fill_forward(f);
    hgs_forward_args& a = f.bw.fwd;
    a.out_color = f.color.data_ptr<float>(), a.radii = P ? f.radii.data_ptr<int32_t>() : nullptr;
    a.visible = (f.s.with_visibility && P) ? (uint8_t*)f.visible.data_ptr<bool>() : nullptr;
    const int64_t M = a.M, M2 = a.seg2.M;
    if (needs_grad && P > 0) {
        f.gl = GradLayout(P1, M, P2, M2);
        f.slab = at::empty({f.gl.total}, fopts);
        point_at_grads(f.bw, f.slab.data_ptr<float>(), f.gl, M, M2);
        a.grad_accum_to_zero = f.bw.grad_accum;
    }
    f.key = std::make_tuple((int)dev.index(), P, H, W);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_hints.find(f.key);
        if (it != g_hints.end()) it->second.stamp = ++g_hint_clock;   // (used: not the next one to go)
        // a shape without history is assumed sparse: the library then allocates the checkpoint buffer only if it is
        // (... unless the shape's last frame said it leaves none: hgs_forward_state.ckpt_slots_used < 0)
        a.backward_checkpoints = (needs_grad && P > 0 && g_use_ckpt && (it == g_hints.end() || ((it->second.sparse || it->second.has_long) && it->second.ckpt_used >= 0))) ? 1 : 0;
        if (g_use_hint && it != g_hints.end()) {
            a.binning_capacity_hint = round_capacity(it->second.n);
            a.expect_no_long_tiles = it->second.has_long ? 0 : 1;
            if (a.backward_checkpoints && it->second.ckpt_used > 0) a.ckpt_slots_hint = round_ckpt_slots(it->second.ckpt_used);
        }
    }
    f.deferred = defer && a.binning_capacity_hint > 0 && P > 0;
    a.defer_n = f.deferred ? 1 : 0;
    if (P > 0) {
        // pre-sized scratch, no allocation callbacks: geom | image | binning(hint) [| checkpoints(hint)]
        const size_t g = align256(hgs_geom_bytes((int32_t)P, (int32_t)H, (int32_t)W)), im = align256(hgs_image_bytes((int32_t)H, (int32_t)W));
        const size_t b = a.binning_capacity_hint > 0 ? align256(hgs_binning_bytes(a.binning_capacity_hint, (int32_t)H, (int32_t)W)) : 0;
        const size_t ck = !(b && a.backward_checkpoints) ? 0
                          : align256(a.ckpt_slots_hint > 0 ? hgs_ckpt_bytes_for_slots(a.ckpt_slots_hint) : hgs_ckpt_bytes(a.binning_capacity_hint, (int32_t)H, (int32_t)W));
        // per frame when backward will need it, else the stream's arena
        f.scratch = needs_grad ? at::empty({(int64_t)(g + im + b + ck)}, bopts) : arena_for((int)dev.index(), (void*)stream, g + im + b + ck, bopts);
        char* base = (char*)f.scratch.data_ptr();
        a.scratch[HGS_BUF_GEOM] = base, a.scratch_bytes[HGS_BUF_GEOM] = g;
        a.scratch[HGS_BUF_IMAGE] = base + g, a.scratch_bytes[HGS_BUF_IMAGE] = im;
        if (b) a.scratch[HGS_BUF_BINNING] = base + g + im, a.scratch_bytes[HGS_BUF_BINNING] = b;
        if (ck) a.scratch[HGS_BUF_CKPT] = base + g + im + b, a.scratch_bytes[HGS_BUF_CKPT] = ck;
    }
    AllocCtx actx{&f.keep, bopts};
    if (before_wait) a.before_wait = before_wait_cb, a.before_wait_ctx = before_wait;
    const int64_t n = hgs_rasterize_forward(&a, alloc_cb, &actx, &f.bw.state, (void*)stream);
    a.before_wait = nullptr, a.before_wait_ctx = nullptr;   // (the argument block lives on in the node: for the backward)
    if (n < 0) raise_last("rasterize_gaussians");
}

// N of the frame (a deferred frame is waited for now -- its scan ran long ago -- and run again, waiting, if it did not fit its
// binning buffer), the shape's record, this thread's "last frame"
void finish_frame(Frame& f, hipStream_t stream)
{
    hgs_forward_args& a = f.bw.fwd;
    int64_t n = f.bw.state.num_rendered;
    if (f.deferred) {
        n = hgs_forward_poll(&f.bw.state, 1, (void*)stream);
        if (n == HGS_ERR_OVERFLOW || n == HGS_ERR_EXPIRED) {
            TORCH_CHECK(std::string(hgs_last_error()).find("2^32") == std::string::npos, "rasterize_gaussians: ", hgs_last_error());
            a.defer_n = 0, a.binning_capacity_hint = 0, a.ckpt_slots_hint = 0, a.scratch[HGS_BUF_BINNING] = nullptr, a.scratch[HGS_BUF_CKPT] = nullptr;
            AllocCtx actx{&f.keep, at::TensorOptions().dtype(at::kByte).device(f.a.means3D.device())};
            n = hgs_rasterize_forward(&a, alloc_cb, &actx, &f.bw.state, (void*)stream);
        }
        if (n < 0) raise_last("rasterize_gaussians (deferred frame)");
        f.deferred = false;
    }
    if (f.P1() + f.P2() == 0) n = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        remember_hint(f.key, Hint{n, f.bw.state.has_long_tiles != 0, f.bw.state.sparse_frame != 0, f.bw.state.ckpt_slots_used});
    }
    t_last_n = n, t_last_capacity = f.bw.state.binning_capacity;
    t_last_long = f.bw.state.has_long_tiles != 0, t_last_sparse = f.bw.state.sparse_frame != 0;
    t_last_ckpt_bytes = (int64_t)f.bw.state.ckpt_bytes, t_last_ckpt_used = f.bw.state.ckpt_slots_used;
}

// ------------------------------------------------------------------------------------------------ the autograd node
// Inputs (= next edges), in this order: means3D, means2D, sh, colors, opacities, scales, rotations, cov3D of the first set, then
// means3D, sh, colors, opacities, scales, rotations, cov3D of the second.  Outputs: the image of frame 0 [, the image of frame 1].
// Frame 1, when present, renders the FIRST set alone (the human-only render next to the joint one).
struct RasterNode : public torch::autograd::Node {
    Frame f[2];
    int nframes = 1;
    bool released = false;
    std::vector<SavedVariable> saved;   // the differentiable inputs: unpacking them runs autograd's in-place-modification check

    std::string name() const override { return "HgsRasterizeBackward"; }

    void release_variables() override
    {
        std::lock_guard<std::mutex> lk(mutex_);
        released = true;
        saved.clear();
        for (Frame& fr : f) {
            fr.a = Set{}, fr.b = Set{};
            fr.scratch = Tensor(), fr.slab = Tensor(), fr.keep.clear();
        }
    }

    // A second backward through the node (retain_graph): the frame gets a fresh slab with a zeroed accumulator.  Allocation and fill
    // run on the CALLER's stream -- so this is called for every frame of the node BEFORE the side stream is fenced behind the caller's:
    // the human-only frame's blend backward adds atomically into its accumulator on the side stream, and must be ordered after the
    // fill (and after whatever the caching allocator's block still had pending on the caller's stream).
    void refresh_slab(Frame& fr)
    {
        if (fr.fresh) return;
        fr.slab = at::empty({fr.gl.total}, fr.a.means3D.options());
        fr.slab.narrow(0, 0, std::max<int64_t>(fr.gl.off[1], 1)).zero_();
        point_at_grads(fr.bw, fr.slab.data_ptr<float>(), fr.gl, fr.bw.fwd.M, fr.bw.fwd.seg2.M);
        fr.fresh = true;   // (as good as the forward's)
    }

    // one frame's backward on `stream`; `add_from`: the frame whose first-set gradients are added in (ready once `wait` has fired)
    void run_backward(Frame& fr, const Tensor& g_color, hipStream_t stream, const Frame* add_from, hipEvent_t wait)
    {
        hgs_backward_args& bw = fr.bw;
        TORCH_INTERNAL_ASSERT(fr.fresh, "refresh_slab() goes first");
        fr.fresh = false;
        bw.dL_dout_color = g_color.data_ptr<float>();
        bw.flags = g_upstream_scale_grad ? HGS_BWD_UPSTREAM_SCALE_GRAD : 0u;
        bw.add_dL_dopacity = bw.add_dL_dcolors = bw.add_dL_dmeans3D = bw.add_dL_dcov3D = bw.add_dL_dsh = bw.add_dL_dscales = bw.add_dL_drotations = nullptr;
        bw.wait_before_per_gaussian = nullptr;
        if (add_from) {
            const hgs_backward_args& o = add_from->bw;
            bw.add_dL_dopacity = o.dL_dopacity, bw.add_dL_dcolors = o.dL_dcolors, bw.add_dL_dmeans3D = o.dL_dmeans3D, bw.add_dL_dcov3D = o.dL_dcov3D;
            bw.add_dL_dsh = o.dL_dsh, bw.add_dL_dscales = o.dL_dscales, bw.add_dL_drotations = o.dL_drotations;
            bw.wait_before_per_gaussian = (void*)wait;
        }
        if (hgs_rasterize_backward(&bw, (void*)stream) < 0) raise_last("rasterize_gaussians_backward");
    }

    variable_list apply(variable_list&& grads) override
    {
        std::lock_guard<std::mutex> lk(mutex_);
        variable_list out(15);
        TORCH_CHECK(!released, "Trying to backward through the rasterizer a second time (or after its buffers have been freed). "
                               "Specify retain_graph=True if you need to backward through the graph a second time.");
        for (auto& sv : saved) (void)sv.unpack(shared_from_this());   // raises if an input was modified in place since the forward
        Frame& J = f[0];
        const int64_t P1 = J.P1(), P2 = J.P2(), P = P1 + P2;
        const bool g0 = grads.size() > 0 && grads[0].defined(), g1 = nframes > 1 && grads.size() > 1 && grads[1].defined();
        if (!g0 && !g1) return out;   // the images did not take part in the loss
        if (P == 0) {   // nothing was rendered (so there is no second set either): empty gradients of the inputs' shapes
            out[0] = at::zeros_like(J.a.means3D), out[1] = at::zeros_like(J.a.means3D);
            const Tensor* firsts[6] = {&J.a.sh, &J.a.colors, &J.a.opac, &J.a.scales, &J.a.rot, &J.a.cov};
            for (int k = 0; k < 6; ++k)
                if (firsts[k]->defined()) out[k + 2] = at::zeros_like(*firsts[k]);
            return out;
        }
        at::AutoGradMode no_grad(false);
        const auto dev = J.a.means3D.device();
        c10::DeviceGuard guard(dev);
        const hipStream_t main = c10::hip::getCurrentHIPStream(dev.index()).stream();
        Frame* src = &J;   // whose slab the first set's (and the joint viewspace) gradients are returned from
        if (g0 && g1) {
            // the human-only frame on the side stream under the joint frame's blend backward; its gradients of the human tensors
            // are added by the joint frame's per-Gaussian kernel, which alone waits for them
            Tensor ga = f32c(grads[0]), gb = f32c(grads[1]);
            Events& ev = events_for((int)dev.index());
            const hipStream_t side = side_stream((int)dev.index()).stream();
            refresh_slab(f[1]), refresh_slab(J);           // (a second backward: both slabs are made and zeroed on `main`, in front of the fence)
            fence(ev.a, main, side);                       // dL/dimage was produced on the caller's stream
            run_backward(f[1], gb, side, nullptr, nullptr);
            TORCH_CHECK(hipEventRecord(ev.b, side) == hipSuccess, "hipEventRecord failed");
            run_backward(J, ga, main, &f[1], ev.b);        // (main has waited for the side stream when this returns)
        } else if (g0) {
            refresh_slab(J);
            run_backward(J, f32c(grads[0]), main, nullptr, nullptr);
        } else {
            refresh_slab(f[1]);
            run_backward(f[1], f32c(grads[1]), main, nullptr, nullptr);
            src = &f[1];
        }
        const hgs_backward_args& bw = src->bw;
        const GradLayout& gl = src->gl;
        const Tensor& slab = src->slab;
        const int64_t M = bw.fwd.M;
        auto view_of = [&](int k, at::IntArrayRef shape) { return slab.narrow(0, gl.off[k], gl.size[k]).view(shape); };
        const Set& a = src->a;
        out[0] = view_of(4, {P1, 3});                                                // means3D
        if (src == &J) out[1] = view_of(1, {P, 3});                                  // means2D (the viewspace gradient sink, both sets)
        if (a.sh.defined()) out[2] = view_of(6, {P1, M, 3});
        if (a.colors.defined()) out[3] = view_of(3, {P1, 3});
        out[4] = view_of(2, {P1, 1});
        if (a.scales.defined()) out[5] = view_of(7, {P1, 3});
        if (a.rot.defined()) out[6] = view_of(8, {P1, 4});
        if (a.cov.defined()) out[7] = view_of(5, {P1, 6});
        if (P2 > 0 && src == &J) {   // the second set's gradients, written in place by the library
            const Set& b = J.b;
            const int64_t M2 = bw.fwd.seg2.M;
            out[8] = view_of(11, {P2, 3});
            if (b.sh.defined()) out[9] = view_of(13, {P2, M2, 3});
            if (b.colors.defined()) out[10] = view_of(10, {P2, 3});
            out[11] = view_of(9, {P2, 1});
            if (b.scales.defined()) out[12] = view_of(14, {P2, 3});
            if (b.rot.defined()) out[13] = view_of(15, {P2, 4});
            if (b.cov.defined()) out[14] = view_of(12, {P2, 6});
        }
        return out;
    }
};

bool any_requires_grad(std::initializer_list<const Tensor*> ts)
{
    if (!at::GradMode::is_enabled()) return false;
    for (const Tensor* t : ts)
        if (t->defined() && t->requires_grad()) return true;
    return false;
}

Settings make_settings(const Tensor& means3D, const Tensor& bg, const Tensor& view, const Tensor& proj, const Tensor& campos, int64_t H, int64_t W,
                       double tanfovx, double tanfovy, double mod, int64_t degree, bool prefiltered, bool debug, bool clamp_output, bool with_visibility)
{
    const auto dev = means3D.device();
    auto on_dev = [&](const Tensor& t) { return f32c(t.device() == dev ? t : t.to(dev)); };
    return Settings{on_dev(bg), on_dev(view), on_dev(proj), on_dev(campos), H, W, degree, tanfovx, tanfovy, mod, prefiltered, debug, clamp_output, with_visibility};
}

// Attach the node: edges to the inputs' gradient functions, the saved inputs, the images as its outputs
void attach(const std::shared_ptr<RasterNode>& node, const Tensor (&inputs)[15])
{
    torch::autograd::edge_list edges;
    edges.reserve(15);
    for (const Tensor& t : inputs) {
        if (t.defined() && t.requires_grad()) {
            edges.push_back(torch::autograd::impl::gradient_edge(t));
            node->saved.emplace_back(t, false);
        } else {
            edges.emplace_back();
        }
    }
    node->set_next_edges(std::move(edges));
    for (int k = 0; k < node->nframes; ++k) {
        torch::autograd::create_gradient_edge(node->f[k].color, node);
        // (the image now owns the node: the node must not own the image -- a reference cycle would keep both, and the frame's
        //  scratch, alive for ever.  radii / visible stay: the backward reads radii, and they do not point back at the node.)
        node->f[k].color = Tensor();
    }
}

// ------------------------------------------------------------------------------------------------ entry points
// The module API: GaussianRasterizer.forward's arguments, upstream's two return values [+ the visibility filter]
std::vector<Tensor> rasterize(Tensor means3D, Tensor means2D, Tensor sh, Tensor colors, Tensor opac, Tensor scales, Tensor rot,
                              Tensor cov, Tensor bg, Tensor view, Tensor proj, Tensor campos, int64_t H, int64_t W,
                              double tanfovx, double tanfovy, double mod, int64_t degree, bool prefiltered, bool debug,
                              bool clamp_output, std::vector<Tensor> second, bool with_visibility)
{
    // `second`: nothing, or the second set's (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    TORCH_CHECK(second.empty() || second.size() == 7, "the second set of Gaussians is a list of seven tensors");
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X): `means3D` must live on the GPU (HIP device); there is no CPU fallback");
    if (second.empty()) second.assign(7, Tensor());
    const bool needs_grad = any_requires_grad({&means3D, &means2D, &sh, &colors, &opac, &scales, &rot, &cov, &second[0], &second[1], &second[2],
                                               &second[3], &second[4], &second[5], &second[6]});
    auto node = std::shared_ptr<RasterNode>(new RasterNode(), torch::autograd::deleteNode);
    Frame& f = node->f[0];
    {
        at::AutoGradMode no_grad(false);
        const auto dev = means3D.device();
        c10::DeviceGuard guard(dev);
        f.a = make_set(means3D, sh, colors, opac, scales, rot, cov);
        if (!f.a.means3D.defined()) f.a.means3D = means3D.to(at::kFloat).reshape({0, 3});
        TORCH_CHECK(f.a.means3D.dim() == 2 && f.a.means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
        f.b = make_set(second[0], second[1], second[2], second[3], second[4], second[5], second[6]);
        f.s = make_settings(means3D, bg, view, proj, campos, H, W, tanfovx, tanfovy, mod, degree, prefiltered, debug, clamp_output, with_visibility);
        const hipStream_t st = c10::hip::getCurrentHIPStream(dev.index()).stream();
        Tensor color, radii, visible;
        BeforeWait bw;
        bw.fn = [&] {   // (runs inside start_frame, before its wait for N -- or below, when the frame had nothing to wait for)
            color = f.color, radii = f.radii, visible = f.visible;
            if (needs_grad) {
                at::AutoGradMode grad_mode(true);
                const Tensor inputs[15] = {means3D, means2D, sh, colors, opac, scales, rot, cov, second[0], second[1], second[2], second[3], second[4], second[5], second[6]};
                attach(node, inputs);
            }
        };
        start_frame(f, needs_grad, st, false, &bw);
        before_wait_cb(&bw);
        finish_frame(f, st);
        if (with_visibility) return {color, radii, visible};
        return {color, radii};
    }
}

// The whole of render() (/root/reference/hugs/renderer/gs_renderer.py:103-161) as one call: the zero viewspace leaf (:107-113),
// tanfov in double (:116-117), `feats.ndim == 2` = precomputed colours (:119-123), the rasterization with the clamp fused (:153)
// and the visibility filter (:159).  `second`: nothing, or the second model's (means3D, feats, opacity, scales, rotations).
// -> (image, radii, visibility_filter, viewspace_points)
std::vector<Tensor> render(Tensor means3D, Tensor feats, Tensor opacity, Tensor scales, Tensor rotations, std::vector<Tensor> second, Tensor bg, Tensor view,
                           Tensor proj, Tensor campos, int64_t H, int64_t W, double fovx, double fovy, double mod, int64_t degree)
{
    TORCH_CHECK(second.empty() || second.size() == 5, "the second model is a list of five tensors");
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X): `means3D` must live on the GPU (HIP device); there is no CPU fallback");
    const bool is_rgb = feats.dim() == 2;
    const Tensor none;
    if (second.empty()) second.assign(5, Tensor());
    const int64_t rows = means3D.size(0) + (second[0].defined() ? second[0].size(0) : 0);
    Tensor viewspace = viewspace_zeros(rows, at::TensorOptions().dtype(means3D.scalar_type()).device(means3D.device()));
    auto out = rasterize(means3D, viewspace, is_rgb ? none : feats, is_rgb ? feats : none, opacity, scales, rotations, none, bg, view, proj, campos, H, W,
                         std::tan(fovx * 0.5), std::tan(fovy * 0.5), mod, degree, false, false, true,
                         second[0].defined() ? std::vector<Tensor>{second[0], is_rgb ? none : second[1], is_rgb ? second[1] : none, second[2], second[3], second[4], none}
                                             : std::vector<Tensor>{},
                         true);
    return {out[0], out[1], out[2], viewspace};
}

// The two renders of one HUGS training step (render_human_scene with render_mode="human_scene", render_human_separate=True:
// gs_renderer.py:56 and :69) as one call and ONE autograd node.  human / scene: (means3D, feats, opacity, scales, rotations).
// -> (image, radii, visibility_filter, viewspace_points, human_img, human_radii, human_visibility_filter)
std::vector<Tensor> render_pair(std::vector<Tensor> human, std::vector<Tensor> scene, Tensor bg, Tensor human_bg, Tensor view, Tensor proj, Tensor campos,
                                int64_t H, int64_t W, double fovx, double fovy, double mod, int64_t degree)
{
    TORCH_CHECK(human.size() == 5 && scene.size() == 5, "human and scene are lists of five tensors");
    TORCH_CHECK(human[0].is_cuda(), "diff_gaussian_rasterization (MI355X): `means3D` must live on the GPU (HIP device); there is no CPU fallback");
    TORCH_CHECK(human[0].size(0) > 0 && scene[0].size(0) > 0 && human[1].dim() == scene[1].dim(), "render_pair: two non-empty models with the same kind of features");
    const bool is_rgb = human[1].dim() == 2;
    const Tensor none;
    const auto dev = human[0].device();
    const int64_t rows = human[0].size(0) + scene[0].size(0);
    Tensor viewspace = viewspace_zeros(rows, at::TensorOptions().dtype(human[0].scalar_type()).device(dev));
    const bool needs_grad = any_requires_grad({&human[0], &human[1], &human[2], &human[3], &human[4], &scene[0], &scene[1], &scene[2], &scene[3], &scene[4]}) ||
                            at::GradMode::is_enabled();   // (the viewspace leaf requires grad)
    auto node = std::shared_ptr<RasterNode>(new RasterNode(), torch::autograd::deleteNode);
    node->nframes = 2;
    Frame &J = node->f[0], &Hh = node->f[1];
    {
        at::AutoGradMode no_grad(false);
        c10::DeviceGuard guard(dev);
        const double tx = std::tan(fovx * 0.5), ty = std::tan(fovy * 0.5);
        J.a = make_set(human[0], is_rgb ? none : human[1], is_rgb ? human[1] : none, human[2], human[3], human[4], none);
        J.b = make_set(scene[0], is_rgb ? none : scene[1], is_rgb ? scene[1] : none, scene[2], scene[3], scene[4], none);
        J.s = make_settings(human[0], bg, view, proj, campos, H, W, tx, ty, mod, degree, false, false, true, true);
        Hh.a = J.a;
        Hh.s = J.s;
        Hh.s.bg = human_bg.defined() ? f32c(human_bg.device() == dev ? human_bg : human_bg.to(dev)) : J.s.bg;
        const hipStream_t main = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const hipStream_t side = side_stream((int)dev.index()).stream();
        Events& ev = events_for((int)dev.index());
        // the human-only frame goes first, on the side stream, without a wait for its N: its latency-bound binning runs under the
        // joint frame's (forward and, through the node, backward); outputs and scratch of both are allocated on the caller's stream
        fence(ev.a, main, side);
        start_frame(Hh, needs_grad, side, true);
        std::vector<Tensor> result;
        BeforeWait bw;
        bw.fn = [&] {   // (inside the joint frame's forward, before its wait for N)
            result = {J.color, J.radii, J.visible, viewspace, Hh.color, Hh.radii, Hh.visible};
            if (needs_grad) {
                at::AutoGradMode grad_mode(true);
                const Tensor inputs[15] = {human[0], viewspace, is_rgb ? none : human[1], is_rgb ? human[1] : none, human[2], human[3], human[4], none,
                                           scene[0], is_rgb ? none : scene[1], is_rgb ? scene[1] : none, scene[2], scene[3], scene[4], none};
                attach(node, inputs);
            }
        };
        try {
            start_frame(J, needs_grad, main, false, &bw);
            before_wait_cb(&bw);
            finish_frame(Hh, side);
            finish_frame(J, main);
        } catch (...) {
            // (an argument error of the joint frame, a failed allocation: the human-only frame may still be running on the side
            //  stream into buffers that are about to be released -- the caller's stream must not reuse them before it is done)
            (void)hipEventRecord(ev.b, side);
            (void)hipStreamWaitEvent(main, ev.b, 0);
            throw;
        }
        fence(ev.b, side, main);   // what follows on the caller's stream sees both images
        return result;
    }
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize", &rasterize, "forward of the differentiable Gaussian rasterizer (C++ autograd node over the C ABI)",
          py::arg("means3D"), py::arg("means2D"), py::arg("sh"), py::arg("colors"), py::arg("opac"), py::arg("scales"), py::arg("rot"),
          py::arg("cov"), py::arg("bg"), py::arg("view"), py::arg("proj"), py::arg("campos"), py::arg("H"), py::arg("W"),
          py::arg("tanfovx"), py::arg("tanfovy"), py::arg("mod"), py::arg("degree"), py::arg("prefiltered"), py::arg("debug"),
          py::arg("clamp_output"), py::arg("second") = std::vector<Tensor>(), py::arg("with_visibility") = false);
    m.def("render", &render, "render() of the HUGS renderer as one call: (image, radii, visibility_filter, viewspace_points)",
          py::arg("means3D"), py::arg("feats"), py::arg("opacity"), py::arg("scales"), py::arg("rotations"), py::arg("second"), py::arg("bg"),
          py::arg("view"), py::arg("proj"), py::arg("campos"), py::arg("H"), py::arg("W"), py::arg("fovx"), py::arg("fovy"), py::arg("mod"),
          py::arg("degree"));
    m.def("render_pair", &render_pair,
          "the joint human+scene render and the separate human-only render of one training step as one call and one autograd node: "
          "(image, radii, visibility_filter, viewspace_points, human_img, human_radii, human_visibility_filter)",
          py::arg("human"), py::arg("scene"), py::arg("bg"), py::arg("human_bg"), py::arg("view"), py::arg("proj"), py::arg("campos"), py::arg("H"),
          py::arg("W"), py::arg("fovx"), py::arg("fovy"), py::arg("mod"), py::arg("degree"));
    m.def("abi_version", [] { return (int)hgs_abi_version(); });
    m.def("last_frame_info", [] { return std::make_tuple(t_last_n, t_last_capacity, t_last_long, t_last_sparse); },
          "(N, binning capacity, has long tiles, sparse) of this thread's last forward");
    m.def("last_ckpt_info", [] { return std::make_tuple(t_last_ckpt_bytes, t_last_ckpt_used); },
          "(bytes of the checkpoint buffer, checkpoint slots used) of this thread's last forward");
    m.def("set_hint", [](int dev, int64_t P, int64_t H, int64_t W, int64_t n, bool has_long, bool sparse, int64_t ckpt_used) {
        std::lock_guard<std::mutex> lk(g_mu);
        remember_hint(std::make_tuple(dev, P, H, W), Hint{n, has_long, sparse, ckpt_used});
    }, py::arg("dev"), py::arg("P"), py::arg("H"), py::arg("W"), py::arg("n"), py::arg("has_long"), py::arg("sparse") = true, py::arg("ckpt_used") = 0);
    m.def("get_hint", [](int dev, int64_t P, int64_t H, int64_t W) -> py::object {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_hints.find(std::make_tuple(dev, P, H, W));
        if (it == g_hints.end()) return py::none();
        return py::make_tuple(it->second.n, it->second.has_long, it->second.sparse, it->second.ckpt_used);
    }, "(N, has long tiles, sparse, checkpoint slots used) of the last frame of this shape, or None");
    m.def("clear_hints", [] { std::lock_guard<std::mutex> lk(g_mu); g_hints.clear(); });
    m.def("use_hints", [](bool on) { g_use_hint = on; });
    m.def("use_checkpoints", [](bool on) { g_use_ckpt = on; });
    m.def("set_upstream_scale_grad", [](bool on) { g_upstream_scale_grad = on; });
}
