// The PyTorch binding of the C ABI (include/hgs_rasterizer.h) as a C++ autograd node: what BASELINE.json's north_star calls
// "a thin C-ABI PyTorch extension".  It does exactly what ml-hugs_amd/diff_gaussian_rasterization/__init__.py's ctypes +
// Python autograd.Function path does -- same library calls, same scratch and hint policy -- without the interpreter in
// the per-frame path: on small frames (the 6 890-Gaussian SMPL template at 512x512) the rasterizer's kernels take ~120 us
// per forward+backward and the Python binding ~290 us of host time.  Host-only code: compiled with g++ against the torch
// headers (tools are in __graft_entry__.build()); no kernels here, PyTorch supplies device memory, streams and autograd.
//
// Mirrors upstream's pybind entry points used at /root/reference/hugs/renderer/gs_renderer.py:144-152 (rasterize_gaussians
// / rasterize_gaussians_backward behind the module's autograd.Function).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <map>
#include <vector>
#include <mutex>
#include <thread>
#include <tuple>

#include "hgs_rasterizer.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Hint { int64_t n; bool has_long; bool sparse; uint64_t stamp = 0; };
uint64_t g_hint_clock = 0;
constexpr size_t HINT_SHAPES = 256;   // shapes remembered; the least recently used go first (round 3 cleared the table at 256)
std::mutex g_mu;
std::map<std::tuple<int, int64_t, int64_t, int64_t>, Hint> g_hints;   // (device, P, H, W) -> previous frame of this shape
// (caller holds g_mu)
void remember_hint(const std::tuple<int, int64_t, int64_t, int64_t>& key, Hint h)
{
    h.stamp = ++g_hint_clock;
    g_hints[key] = h;
    while (g_hints.size() > HINT_SHAPES) {   // densification changes P all the time: drop the least recently used quarter
        std::vector<std::pair<uint64_t, std::tuple<int, int64_t, int64_t, int64_t>>> by_age;
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

void fill_segment(hgs_segment& g, const Tensor& means3D, const Tensor& sh, const Tensor& colors, const Tensor& opac,
                  const Tensor& scales, const Tensor& rot, const Tensor& cov)
{
    memset(&g, 0, sizeof g);
    if (!means3D.defined() || means3D.numel() == 0) return;
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "second: means3D must have dimensions (num_points, 3)");
    check_segment_tensor("shs", sh, means3D, 0, true);
    check_segment_tensor("colors_precomp", colors, means3D, 3);
    check_segment_tensor("opacities", opac, means3D, 1);
    check_segment_tensor("scales", scales, means3D, 3);
    check_segment_tensor("rotations", rot, means3D, 4);
    check_segment_tensor("cov3D_precomp", cov, means3D, 6);
    g.P = (int32_t)means3D.size(0);
    g.M = sh.defined() && sh.numel() ? (int32_t)sh.size(1) : 0;
    g.means3D = fptr(means3D), g.shs = fptr(sh), g.colors_precomp = fptr(colors), g.opacities = fptr(opac);
    g.scales = fptr(scales), g.rotations = fptr(rot), g.cov3D_precomp = fptr(cov);
}

void fill_forward(hgs_forward_args& a, const Tensor& means3D, const Tensor& sh, const Tensor& colors, const Tensor& opac,
                  const Tensor& scales, const Tensor& rot, const Tensor& cov, const Tensor& bg, const Tensor& view,
                  const Tensor& proj, const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double mod,
                  int64_t degree, bool prefiltered, bool debug, bool clamp_output)
{
    memset(&a, 0, sizeof a);
    a.s.image_height = (int32_t)H, a.s.image_width = (int32_t)W;
    a.s.tanfovx = (float)tanfovx, a.s.tanfovy = (float)tanfovy;
    a.s.bg = fptr(bg), a.s.viewmatrix = fptr(view), a.s.projmatrix = fptr(proj), a.s.campos = fptr(campos);
    a.s.scale_modifier = (float)mod, a.s.sh_degree = (int32_t)degree, a.s.prefiltered = prefiltered, a.s.debug = debug;
    a.P = (int32_t)means3D.size(0);
    a.M = sh.defined() && sh.numel() ? (int32_t)sh.size(1) : 0;
    a.means3D = fptr(means3D), a.shs = fptr(sh), a.colors_precomp = fptr(colors), a.opacities = fptr(opac);
    a.scales = fptr(scales), a.rotations = fptr(rot), a.cov3D_precomp = fptr(cov);
    a.clamp_output = clamp_output ? 1 : 0;
}

class Rasterize : public torch::autograd::Function<Rasterize> {
public:
    static variable_list forward(AutogradContext* ctx, Tensor means3D_, Tensor means2D, Tensor sh_, Tensor colors_,
                                 Tensor opac_, Tensor scales_, Tensor rot_, Tensor cov_, Tensor bg_, Tensor view_,
                                 Tensor proj_, Tensor campos_, int64_t H, int64_t W, double tanfovx, double tanfovy,
                                 double mod, int64_t degree, bool prefiltered, bool debug, bool clamp_output, bool needs_grad,
                                 Tensor means3D_b_, Tensor sh_b_, Tensor colors_b_, Tensor opac_b_, Tensor scales_b_, Tensor rot_b_,
                                 Tensor cov_b_, bool with_visibility)
    {
        TORCH_CHECK(means3D_.is_cuda(), "diff_gaussian_rasterization (MI355X): `means3D` must live on the GPU (HIP device); there is no CPU fallback");
        const auto dev = means3D_.device();
        Tensor means3D = means3D_.numel() ? f32c(means3D_) : means3D_.to(at::kFloat).reshape({0, 3});
        TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
        Tensor sh = f32c(sh_), colors = f32c(colors_), opac = f32c(opac_), scales = f32c(scales_), rot = f32c(rot_), cov = f32c(cov_);
        auto on_dev = [&](const Tensor& t) { return f32c(t.device() == dev ? t : t.to(dev)); };
        Tensor bg = on_dev(bg_), view = on_dev(view_), proj = on_dev(proj_), campos = on_dev(campos_);
        // the optional second set of Gaussians (hgs_segment): rendered behind the first in index order, nothing concatenated
        Tensor means3D_b = f32c(means3D_b_), sh_b = f32c(sh_b_), colors_b = f32c(colors_b_), opac_b = f32c(opac_b_),
               scales_b = f32c(scales_b_), rot_b = f32c(rot_b_), cov_b = f32c(cov_b_);
        const int64_t P1 = means3D.size(0), P2 = means3D_b.defined() ? means3D_b.size(0) : 0;
        TORCH_CHECK(P2 == 0 || P1 > 0, "a second set of Gaussians needs a non-empty first one");
        TORCH_CHECK(P2 == 0 || means3D_b.device() == dev, "second: means3D is on ", means3D_b.device(), ", the first set of Gaussians on ", dev);
        TORCH_CHECK(P2 == 0 || (means3D_b.dim() == 2 && means3D_b.size(1) == 3), "means3D must have dimensions (num_points, 3)");
        const int64_t P = P1 + P2;
        const auto fopts = at::TensorOptions().dtype(at::kFloat).device(dev);
        const auto bopts = at::TensorOptions().dtype(at::kByte).device(dev);
        Tensor color = P == 0 ? at::zeros({3, H, W}, fopts) : at::empty({3, H, W}, fopts);
        Tensor radii = at::empty({P}, fopts.dtype(at::kInt));
        Tensor visible = with_visibility ? at::empty({P}, fopts.dtype(at::kBool)) : Tensor();   // `radii > 0`, written with radii

        hgs_backward_args bw;
        memset(&bw, 0, sizeof bw);
        hgs_forward_args& a = bw.fwd;
        fill_forward(a, means3D, sh, colors, opac, scales, rot, cov, bg, view, proj, campos, H, W, tanfovx, tanfovy, mod, degree,
                     prefiltered, debug, clamp_output);
        fill_segment(a.seg2, means3D_b, sh_b, colors_b, opac_b, scales_b, rot_b, cov_b);
        a.out_color = color.data_ptr<float>(), a.radii = P ? radii.data_ptr<int32_t>() : nullptr;
        a.visible = (with_visibility && P) ? (uint8_t*)visible.data_ptr<bool>() : nullptr;
        const int64_t M = a.M, M2 = a.seg2.M;
        // (needs_grad is decided by the caller: grad mode is off inside forward())
        Tensor slab;
        if (needs_grad && P > 0) {
            GradLayout g(P1, M, P2, M2);
            slab = at::empty({g.total}, fopts);
            point_at_grads(bw, slab.data_ptr<float>(), g, M, M2);
            a.grad_accum_to_zero = bw.grad_accum;
        }
        const auto key = std::make_tuple((int)dev.index(), P, H, W);
        {
            std::lock_guard<std::mutex> lk(g_mu);
            auto it = g_hints.find(key);
            if (it != g_hints.end()) it->second.stamp = ++g_hint_clock;   // (used: not the next one to go)
            // a shape without history is assumed sparse: the library then allocates the checkpoint buffer only if it is
            a.backward_checkpoints = (needs_grad && P > 0 && g_use_ckpt && (it == g_hints.end() || it->second.sparse || it->second.has_long)) ? 1 : 0;
            if (g_use_hint && it != g_hints.end()) {
                a.binning_capacity_hint = round_capacity(it->second.n);
                a.expect_no_long_tiles = it->second.has_long ? 0 : 1;
            }
        }
        std::vector<Tensor> keep;
        AllocCtx actx{&keep, bopts};
        Tensor scratch;
        if (P > 0) {
            // pre-sized scratch, no allocation callbacks: geom | image | binning(hint) [| checkpoints(hint)]
            const size_t g = align256(hgs_geom_bytes((int32_t)P, (int32_t)H, (int32_t)W)), im = align256(hgs_image_bytes((int32_t)H, (int32_t)W));
            const size_t b = a.binning_capacity_hint > 0 ? align256(hgs_binning_bytes(a.binning_capacity_hint, (int32_t)H, (int32_t)W)) : 0;
            const size_t ck = b && a.backward_checkpoints ? align256(hgs_ckpt_bytes(a.binning_capacity_hint, (int32_t)H, (int32_t)W)) : 0;
            // per frame when backward will need it, else the stream's arena
            scratch = needs_grad ? at::empty({(int64_t)(g + im + b + ck)}, bopts)
                                 : arena_for((int)dev.index(), (void*)c10::hip::getCurrentHIPStream(dev.index()).stream(), g + im + b + ck, bopts);
            char* base = (char*)scratch.data_ptr();
            a.scratch[HGS_BUF_GEOM] = base, a.scratch_bytes[HGS_BUF_GEOM] = g;
            a.scratch[HGS_BUF_IMAGE] = base + g, a.scratch_bytes[HGS_BUF_IMAGE] = im;
            if (b) a.scratch[HGS_BUF_BINNING] = base + g + im, a.scratch_bytes[HGS_BUF_BINNING] = b;
            if (ck) a.scratch[HGS_BUF_CKPT] = base + g + im + b, a.scratch_bytes[HGS_BUF_CKPT] = ck;
        }
        int64_t n;
        {
            c10::DeviceGuard guard(dev);
            n = hgs_rasterize_forward(&a, alloc_cb, &actx, &bw.state, (void*)c10::hip::getCurrentHIPStream(dev.index()).stream());
        }
        if (n < 0) raise_last("rasterize_gaussians");
        {
            std::lock_guard<std::mutex> lk(g_mu);
            remember_hint(key, Hint{n, bw.state.has_long_tiles != 0, bw.state.sparse_frame != 0});
        }
        t_last_n = n, t_last_capacity = bw.state.binning_capacity;
        t_last_long = bw.state.has_long_tiles != 0, t_last_sparse = bw.state.sparse_frame != 0;

        if (with_visibility) ctx->mark_non_differentiable({radii, visible});
        else ctx->mark_non_differentiable({radii});
        if (needs_grad) {
            ctx->set_materialize_grads(false);
            ctx->save_for_backward({means3D, sh.defined() ? sh : Tensor(), colors.defined() ? colors : Tensor(),
                                    opac.defined() ? opac : Tensor(), scales.defined() ? scales : Tensor(),
                                    rot.defined() ? rot : Tensor(), cov.defined() ? cov : Tensor(), radii, bg, view, proj, campos,
                                    scratch, slab, means3D_b.defined() ? means3D_b : Tensor(), sh_b.defined() ? sh_b : Tensor(),
                                    colors_b.defined() ? colors_b : Tensor(), opac_b.defined() ? opac_b : Tensor(),
                                    scales_b.defined() ? scales_b : Tensor(), rot_b.defined() ? rot_b : Tensor(),
                                    cov_b.defined() ? cov_b : Tensor()});
            // (buffers the allocation callback handed out -- the binning / checkpoint buffers of an unhinted or under-guessed
            // frame -- stay alive with the node)
            for (const Tensor& t : keep) ctx->saved_data["keep" + std::to_string(&t - keep.data())] = t;
            ctx->saved_data["H"] = H, ctx->saved_data["W"] = W, ctx->saved_data["tx"] = tanfovx, ctx->saved_data["ty"] = tanfovy;
            ctx->saved_data["mod"] = mod, ctx->saved_data["D"] = degree, ctx->saved_data["flags"] = (int64_t)((prefiltered ? 1 : 0) | (debug ? 2 : 0) | (clamp_output ? 4 : 0));
            ctx->saved_data["N"] = n, ctx->saved_data["cap"] = bw.state.binning_capacity;
            ctx->saved_data["sparse"] = (int64_t)bw.state.sparse_frame, ctx->saved_data["long"] = (int64_t)bw.state.has_long_tiles;
            ctx->saved_data["geom"] = (int64_t)(uintptr_t)bw.state.geom, ctx->saved_data["geom_b"] = (int64_t)bw.state.geom_bytes;
            ctx->saved_data["bin"] = (int64_t)(uintptr_t)bw.state.binning, ctx->saved_data["bin_b"] = (int64_t)bw.state.binning_bytes;
            ctx->saved_data["img"] = (int64_t)(uintptr_t)bw.state.image, ctx->saved_data["img_b"] = (int64_t)bw.state.image_bytes;
            ctx->saved_data["ck"] = (int64_t)(uintptr_t)bw.state.ckpt, ctx->saved_data["ck_b"] = (int64_t)bw.state.ckpt_bytes;
            ctx->saved_data["fresh"] = true;
        }
        if (with_visibility) return {color, radii, visible};
        return {color, radii};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        variable_list out(30);
        const Tensor& g_color = grads[0];
        if (!g_color.defined()) return out;   // colour did not take part in the loss
        const auto sv = ctx->get_saved_variables();
        if (sv.empty()) return out;
        const Tensor &means3D = sv[0], &sh = sv[1], &colors = sv[2], &opac = sv[3], &scales = sv[4], &rot = sv[5], &cov = sv[6],
                     &radii = sv[7], &bg = sv[8], &view = sv[9], &proj = sv[10], &campos = sv[11];
        Tensor slab = sv[13];
        const Tensor &means3D_b = sv[14], &sh_b = sv[15], &colors_b = sv[16], &opac_b = sv[17], &scales_b = sv[18], &rot_b = sv[19],
                     &cov_b = sv[20];
        const int64_t P1 = means3D.size(0), P2 = means3D_b.defined() ? means3D_b.size(0) : 0;
        const int64_t P = P1 + P2, H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        if (P == 0) {   // nothing was rendered (so there is no second set either): empty gradients of the inputs' shapes
            out[0] = at::zeros_like(means3D), out[1] = at::zeros_like(means3D);
            for (int k = 1; k <= 6; ++k)
                if (sv[k].defined()) out[k + 1] = at::zeros_like(sv[k]);
            return out;
        }
        const int64_t flags = ctx->saved_data["flags"].toInt();
        hgs_backward_args bw;
        memset(&bw, 0, sizeof bw);
        fill_forward(bw.fwd, means3D, sh, colors, opac, scales, rot, cov, bg, view, proj, campos, H, W, ctx->saved_data["tx"].toDouble(),
                     ctx->saved_data["ty"].toDouble(), ctx->saved_data["mod"].toDouble(), ctx->saved_data["D"].toInt(), flags & 1, flags & 2,
                     flags & 4);
        fill_segment(bw.fwd.seg2, means3D_b, sh_b, colors_b, opac_b, scales_b, rot_b, cov_b);
        bw.fwd.radii = radii.data_ptr<int32_t>();
        const int64_t M = bw.fwd.M, M2 = bw.fwd.seg2.M;
        bw.state.geom = (void*)(uintptr_t)ctx->saved_data["geom"].toInt(), bw.state.geom_bytes = (size_t)ctx->saved_data["geom_b"].toInt();
        bw.state.binning = (void*)(uintptr_t)ctx->saved_data["bin"].toInt(), bw.state.binning_bytes = (size_t)ctx->saved_data["bin_b"].toInt();
        bw.state.image = (void*)(uintptr_t)ctx->saved_data["img"].toInt(), bw.state.image_bytes = (size_t)ctx->saved_data["img_b"].toInt();
        bw.state.ckpt = (void*)(uintptr_t)ctx->saved_data["ck"].toInt(), bw.state.ckpt_bytes = (size_t)ctx->saved_data["ck_b"].toInt();
        bw.state.num_rendered = ctx->saved_data["N"].toInt(), bw.state.binning_capacity = ctx->saved_data["cap"].toInt();
        bw.state.sparse_frame = (int32_t)ctx->saved_data["sparse"].toInt(), bw.state.has_long_tiles = (int32_t)ctx->saved_data["long"].toInt();
        GradLayout gl(P1, M, P2, M2);
        const auto dev = means3D.device();
        if (!ctx->saved_data["fresh"].toBool()) {   // a second backward (retain_graph): a fresh, zeroed slab
            slab = at::empty({gl.total}, means3D.options());
            slab.narrow(0, 0, std::max<int64_t>(gl.off[1], 1)).zero_();
        }
        ctx->saved_data["fresh"] = false;
        point_at_grads(bw, slab.data_ptr<float>(), gl, M, M2);
        Tensor g = f32c(g_color);
        bw.dL_dout_color = g.data_ptr<float>();
        bw.flags = g_upstream_scale_grad ? HGS_BWD_UPSTREAM_SCALE_GRAD : 0u;
        int32_t rc;
        {
            c10::DeviceGuard guard(dev);
            rc = hgs_rasterize_backward(&bw, (void*)c10::hip::getCurrentHIPStream(dev.index()).stream());
        }
        if (rc < 0) raise_last("rasterize_gaussians_backward");
        auto view_of = [&](int k, at::IntArrayRef shape) { return slab.narrow(0, gl.off[k], gl.size[k]).view(shape); };
        out[0] = view_of(4, {P1, 3});                                 // means3D
        out[1] = view_of(1, {P, 3});                                  // means2D (the viewspace gradient sink, both sets)
        if (sh.defined() && sh.numel()) out[2] = view_of(6, {P1, M, 3});
        if (colors.defined() && colors.numel()) out[3] = view_of(3, {P1, 3});
        out[4] = view_of(2, {P1, 1});
        if (scales.defined() && scales.numel()) out[5] = view_of(7, {P1, 3});
        if (rot.defined() && rot.numel()) out[6] = view_of(8, {P1, 4});
        if (cov.defined() && cov.numel()) out[7] = view_of(5, {P1, 6});
        if (P2 > 0) {   // the second set's gradients, written in place by the library
            out[22] = view_of(11, {P2, 3});
            if (sh_b.defined() && sh_b.numel()) out[23] = view_of(13, {P2, M2, 3});
            if (colors_b.defined() && colors_b.numel()) out[24] = view_of(10, {P2, 3});
            out[25] = view_of(9, {P2, 1});
            if (scales_b.defined() && scales_b.numel()) out[26] = view_of(14, {P2, 3});
            if (rot_b.defined() && rot_b.numel()) out[27] = view_of(15, {P2, 4});
            if (cov_b.defined() && cov_b.numel()) out[28] = view_of(12, {P2, 6});
        }
        return out;
    }
};

std::vector<Tensor> rasterize(Tensor means3D, Tensor means2D, Tensor sh, Tensor colors, Tensor opac, Tensor scales, Tensor rot,
                              Tensor cov, Tensor bg, Tensor view, Tensor proj, Tensor campos, int64_t H, int64_t W,
                              double tanfovx, double tanfovy, double mod, int64_t degree, bool prefiltered, bool debug,
                              bool clamp_output, std::vector<Tensor> second, bool with_visibility)
{
    // `second`: nothing, or the second set's (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    TORCH_CHECK(second.empty() || second.size() == 7, "the second set of Gaussians is a list of seven tensors");
    if (second.empty()) second.assign(7, at::empty({0}, means3D.options()));
    bool needs_grad = false;
    if (at::GradMode::is_enabled()) {
        for (const Tensor* t : {&means3D, &means2D, &sh, &colors, &opac, &scales, &rot, &cov})
            needs_grad = needs_grad || (t->defined() && t->requires_grad());
        for (const Tensor& t : second) needs_grad = needs_grad || (t.defined() && t.requires_grad());
    }
    auto r = Rasterize::apply(means3D, means2D, sh, colors, opac, scales, rot, cov, bg, view, proj, campos, H, W, tanfovx, tanfovy,
                              mod, degree, prefiltered, debug, clamp_output, needs_grad, second[0], second[1], second[2], second[3],
                              second[4], second[5], second[6], with_visibility);
    if (with_visibility) return {r[0], r[1], r[2]};
    return {r[0], r[1]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize", &rasterize, "forward of the differentiable Gaussian rasterizer (C++ autograd node over the C ABI)",
          py::arg("means3D"), py::arg("means2D"), py::arg("sh"), py::arg("colors"), py::arg("opac"), py::arg("scales"), py::arg("rot"),
          py::arg("cov"), py::arg("bg"), py::arg("view"), py::arg("proj"), py::arg("campos"), py::arg("H"), py::arg("W"),
          py::arg("tanfovx"), py::arg("tanfovy"), py::arg("mod"), py::arg("degree"), py::arg("prefiltered"), py::arg("debug"),
          py::arg("clamp_output"), py::arg("second") = std::vector<Tensor>(), py::arg("with_visibility") = false);
    m.def("abi_version", [] { return (int)hgs_abi_version(); });
    m.def("last_frame_info", [] { return std::make_tuple(t_last_n, t_last_capacity, t_last_long, t_last_sparse); },
          "(N, binning capacity, has long tiles, sparse) of this thread's last forward");
    m.def("set_hint", [](int dev, int64_t P, int64_t H, int64_t W, int64_t n, bool has_long, bool sparse) {
        std::lock_guard<std::mutex> lk(g_mu);
        remember_hint(std::make_tuple(dev, P, H, W), Hint{n, has_long, sparse});
    }, py::arg("dev"), py::arg("P"), py::arg("H"), py::arg("W"), py::arg("n"), py::arg("has_long"), py::arg("sparse") = true);
    m.def("clear_hints", [] { std::lock_guard<std::mutex> lk(g_mu); g_hints.clear(); });
    m.def("use_hints", [](bool on) { g_use_hint = on; });
    m.def("use_checkpoints", [](bool on) { g_use_ckpt = on; });
    m.def("set_upstream_scale_grad", [](bool on) { g_upstream_scale_grad = on; });
}
