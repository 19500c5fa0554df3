// A torch-free host for the C ABI of include/hgs_rasterizer.h: everything a C/C++ (or cgo / JNI / FFI) caller does.
// Reads a scene from a flat binary file, runs hgs_rasterize_forward + hgs_rasterize_backward with plain hipMalloc'd
// buffers and a hipMalloc allocation callback, and writes the results back for tests/test_c_host.py to compare with
// the oracle.  Built by __graft_entry__.build():  hipcc -O2 raster_host.cpp -L<lib> -lhgs_rasterizer
//
// use_hint = 2 adds what a frame loop does: a DEFERRED frame (defer_n) into caller-provided scratch sized by hgs_*_bytes,
// resolved by hgs_forward_poll -- backward refused before that -- and a deferred frame with too small a capacity, which must
// come back as HGS_ERR_OVERFLOW.
//
// use_hint = 4: the ABI v11 additions from a plain C caller -- a checkpoint buffer sized by ckpt_slots_hint (what the frame before
// used), the before_wait callback, and a second backward whose per-Gaussian kernel ADDS another backward's gradients (add_*) after
// waiting for an event of another stream (wait_before_per_gaussian): every per-input gradient written out is then TWICE the frame's.
//
//   in.bin : int32 P, M, H, W, D, use_hint | float tanfovx, tanfovy, scale_modifier | bg[3] view[16] proj[16] campos[3]
//            means3D[3P] shs[3MP] opacities[P] scales[3P] rotations[4P] dL_dcolor[3HW]
//            (use_hint = 5: cov3D_precomp[6P] in place of scales and rotations; out.bin then ends with dL_dcov3D[6P])
//   out.bin: int64 N | color[3HW] | int32 radii[P] | dL_dmeans3D[3P] dL_dmeans2D[3P] dL_dopacity[P] dL_dsh[3MP]
//            dL_dscales[3P] dL_drotations[4P]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hgs_rasterizer.h"

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)

static int g_before_wait_calls = 0;
static void before_wait_cb(void* ctx) { ++*(int*)ctx; }
static std::vector<void*> g_scratch;
static void* alloc_cb(void*, int, size_t bytes)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    g_scratch.push_back(p);
    return p;
}

template <class T>
static T* to_device(const std::vector<T>& h)
{
    T* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
template <class T>
static T* device_buffer(size_t n)
{
    T* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(T) + 16) != hipSuccess) return nullptr;
    return d;
}
template <class T>
static bool read_vec(FILE* f, std::vector<T>& v, size_t n)
{
    v.resize(n);
    return fread(v.data(), sizeof(T), n, f) == n;
}
template <class T>
static void write_dev(FILE* f, const T* d, size_t n)
{
    std::vector<T> h(n);
    (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    fwrite(h.data(), sizeof(T), n, f);
}

int main(int argc, char** argv)
{
    if (argc != 3) return fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]), 1;
    if (hgs_abi_version() != HGS_ABI_VERSION) return fprintf(stderr, "ABI mismatch\n"), 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return perror(argv[1]), 1;
    int32_t hdr[6];
    float fl[3];
    if (fread(hdr, 4, 6, f) != 6 || fread(fl, 4, 3, f) != 3) return 1;
    const int P = hdr[0], M = hdr[1], H = hdr[2], W = hdr[3], D = hdr[4], use_hint = hdr[5];
    const bool cov_mode = use_hint == 5;   // (mode 5: precomputed 3-D covariances where the scales and rotations would be)
    std::vector<float> bg, view, proj, campos, means, shs, opac, scales, rots, cov, dL;
    if (!read_vec(f, bg, 3) || !read_vec(f, view, 16) || !read_vec(f, proj, 16) || !read_vec(f, campos, 3) ||
        !read_vec(f, means, 3 * (size_t)P) || !read_vec(f, shs, 3 * (size_t)M * P) || !read_vec(f, opac, P) ||
        !(cov_mode ? read_vec(f, cov, 6 * (size_t)P) : (read_vec(f, scales, 3 * (size_t)P) && read_vec(f, rots, 4 * (size_t)P))) ||
        !read_vec(f, dL, 3 * (size_t)H * W))
        return fprintf(stderr, "short input file\n"), 1;
    fclose(f);

    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    hgs_backward_args bw = {};          // embeds the forward arguments and the forward state
    hgs_forward_args& a = bw.fwd;
    a.s.image_height = H, a.s.image_width = W, a.s.tanfovx = fl[0], a.s.tanfovy = fl[1], a.s.scale_modifier = fl[2];
    a.s.bg = to_device(bg), a.s.viewmatrix = to_device(view), a.s.projmatrix = to_device(proj), a.s.campos = to_device(campos);
    a.s.sh_degree = D, a.s.prefiltered = 0, a.s.debug = 0;
    a.P = P, a.M = M;
    a.means3D = to_device(means), a.shs = to_device(shs), a.opacities = to_device(opac);
    if (cov_mode) a.cov3D_precomp = to_device(cov);
    else a.scales = to_device(scales), a.rotations = to_device(rots);
    float* color = device_buffer<float>(3 * (size_t)H * W);
    int32_t* radii = device_buffer<int32_t>(P);
    a.out_color = color, a.radii = radii;
    bw.grad_accum = device_buffer<float>((size_t)P * 12);
    a.grad_accum_to_zero = bw.grad_accum;  // forward zeroes backward's accumulator
    bw.dL_dmeans2D = device_buffer<float>(3 * (size_t)P), bw.dL_dopacity = device_buffer<float>(P);
    bw.dL_dcolors = device_buffer<float>(3 * (size_t)P), bw.dL_dmeans3D = device_buffer<float>(3 * (size_t)P);
    bw.dL_dcov3D = device_buffer<float>(6 * (size_t)P), bw.dL_dsh = device_buffer<float>(3 * (size_t)M * P);
    bw.dL_dscales = device_buffer<float>(3 * (size_t)P), bw.dL_drotations = device_buffer<float>(4 * (size_t)P);
    bw.dL_dout_color = to_device(dL);
    if (use_hint == 3) {
        // the two-segment form (hgs_segment): the same Gaussians handed over as a first set of P1 and a second set of P - P1.
        // Here both live in one allocation, so the second set's inputs and gradients are simply the tails of the arrays --
        // and everything written out below must equal the one-set run.
        const size_t P1 = (size_t)(2 * P) / 5;
        a.P = (int32_t)P1;
        hgs_segment& b = a.seg2;
        b.P = P - (int32_t)P1, b.M = M;
        b.means3D = a.means3D + 3 * P1, b.shs = a.shs + 3 * (size_t)M * P1, b.opacities = a.opacities + P1;
        b.scales = a.scales + 3 * P1, b.rotations = a.rotations + 4 * P1;
        bw.seg2_dL_dopacity = bw.dL_dopacity + P1, bw.seg2_dL_dcolors = bw.dL_dcolors + 3 * P1, bw.seg2_dL_dmeans3D = bw.dL_dmeans3D + 3 * P1;
        bw.seg2_dL_dcov3D = bw.dL_dcov3D + 6 * P1, bw.seg2_dL_dsh = bw.dL_dsh + 3 * (size_t)M * P1;
        bw.seg2_dL_dscales = bw.dL_dscales + 3 * P1, bw.seg2_dL_drotations = bw.dL_drotations + 4 * P1;
    }

    // with a hint the host also offers the checkpoint buffer of the depth-segmented backward (small frames are sparse frames)
    a.backward_checkpoints = use_hint ? 1 : 0;   // (mode 3 too)
    int64_t N = -1;
    for (int frame = 0; frame < (use_hint ? 2 : 1); ++frame) {   // second frame: capacity guessed from the first
        a.binning_capacity_hint = frame == 0 ? 0 : N + N / 8 + 4096;
        N = hgs_rasterize_forward(&a, alloc_cb, nullptr, &bw.state, stream);
        if (N < 0) return fprintf(stderr, "forward: %s\n", hgs_last_error()), 3;
    }
    if (use_hint == 2) {
        const int64_t cap = N + N / 8 + 4096;
        const size_t bytes[HGS_NUM_BUFS] = {hgs_geom_bytes(P, H, W), hgs_binning_bytes(cap, H, W), hgs_image_bytes(H, W), hgs_ckpt_bytes(cap, H, W)};
        for (int k = 0; k < HGS_NUM_BUFS; ++k) {
            a.scratch[k] = alloc_cb(nullptr, k, bytes[k]), a.scratch_bytes[k] = bytes[k];
            if (!a.scratch[k]) return fprintf(stderr, "scratch allocation failed\n"), 2;
        }
        // a deferred frame with too small a capacity: enqueued without complaint, found out by the poll
        a.defer_n = 1, a.binning_capacity_hint = N > 1 ? N / 2 : 1;
        if (hgs_rasterize_forward(&a, alloc_cb, nullptr, &bw.state, stream) != 0 || bw.state.num_rendered != -1)
            return fprintf(stderr, "deferred forward: %s\n", hgs_last_error()), 3;
        if (N > 1 && hgs_forward_poll(&bw.state, 1, stream) != HGS_ERR_OVERFLOW) return fprintf(stderr, "an overflowed deferred frame was not reported\n"), 3;
        // the real one
        const size_t n_scratch = g_scratch.size();
        a.binning_capacity_hint = cap;
        if (hgs_rasterize_forward(&a, alloc_cb, nullptr, &bw.state, stream) != 0 || bw.state.num_rendered != -1)
            return fprintf(stderr, "deferred forward: %s\n", hgs_last_error()), 3;
        if (g_scratch.size() != n_scratch) return fprintf(stderr, "the allocation callback ran although scratch was provided\n"), 3;
        if (hgs_rasterize_backward(&bw, stream) != HGS_ERR_INVALID_ARGUMENT) return fprintf(stderr, "backward accepted an unresolved deferred frame\n"), 3;
        const int64_t n_deferred = hgs_forward_poll(&bw.state, 1, stream);
        if (n_deferred != N || bw.state.num_rendered != N)
            return fprintf(stderr, "poll: %lld, expected %lld (%s)\n", (long long)n_deferred, (long long)N, hgs_last_error()), 3;
        if (hgs_forward_poll(&bw.state, 0, stream) != N) return fprintf(stderr, "a second poll must repeat N\n"), 3;
    }
    if (hgs_rasterize_backward(&bw, stream) < 0) return fprintf(stderr, "backward: %s\n", hgs_last_error()), 3;
    CHECK(hipStreamSynchronize(stream));
    if (use_hint == 4) {
        // keep the frame's gradients, render it again -- the checkpoint buffer laid out for what the frame before used, the callback
        // in front of the wait for N -- and run a backward that adds them to its own
        struct Copy { float** dst; const float* src; size_t n; };
        float *g_op, *g_col, *g_m3, *g_cov, *g_sh, *g_sc, *g_rot;
        const Copy copies[] = {{&g_op, bw.dL_dopacity, (size_t)P}, {&g_col, bw.dL_dcolors, 3 * (size_t)P}, {&g_m3, bw.dL_dmeans3D, 3 * (size_t)P},
                               {&g_cov, bw.dL_dcov3D, 6 * (size_t)P}, {&g_sh, bw.dL_dsh, 3 * (size_t)M * P}, {&g_sc, bw.dL_dscales, 3 * (size_t)P},
                               {&g_rot, bw.dL_drotations, 4 * (size_t)P}};
        for (const Copy& c : copies) {
            *c.dst = device_buffer<float>(c.n);
            CHECK(hipMemcpy(*c.dst, c.src, c.n * sizeof(float), hipMemcpyDeviceToDevice));
        }
        const int64_t used = bw.state.ckpt_slots_used;
        if (!bw.state.ckpt || used <= 0) return fprintf(stderr, "the sparse frame left no checkpoints (slots used %lld)\n", (long long)used), 3;
        a.binning_capacity_hint = N + N / 8 + 4096, a.ckpt_slots_hint = used + 64;
        a.before_wait = before_wait_cb, a.before_wait_ctx = &g_before_wait_calls;
        if (hgs_rasterize_forward(&a, alloc_cb, nullptr, &bw.state, stream) != N) return fprintf(stderr, "forward (slots hint): %s\n", hgs_last_error()), 3;
        if (g_before_wait_calls != 1) return fprintf(stderr, "before_wait ran %d times\n", g_before_wait_calls), 3;
        if (bw.state.ckpt_slots != used + 64 || bw.state.ckpt_bytes != hgs_ckpt_bytes_for_slots(used + 64) || bw.state.ckpt_slots_used != used)
            return fprintf(stderr, "checkpoint buffer: %lld slots, %zu bytes, %lld used\n", (long long)bw.state.ckpt_slots, bw.state.ckpt_bytes, (long long)bw.state.ckpt_slots_used), 3;
        // a guess that is too small is repaired (the frame is run again, exactly sized)
        a.ckpt_slots_hint = used / 2 > 0 ? used / 2 : 1;
        if (hgs_rasterize_forward(&a, alloc_cb, nullptr, &bw.state, stream) != N || bw.state.ckpt_slots < used)
            return fprintf(stderr, "forward (slots guess too small): %s, %lld slots\n", hgs_last_error(), (long long)bw.state.ckpt_slots), 3;
        if (hgs_debug_stat("ckpt_reruns") < 1) return fprintf(stderr, "the too-small checkpoint guess was not counted as a re-run\n"), 3;
        a.before_wait = nullptr;
        hipStream_t other;
        hipEvent_t ready;
        CHECK(hipStreamCreate(&other));
        CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        CHECK(hipEventRecord(ready, other));   // (the "other render" finished long ago; what matters is that the wait is accepted and honoured)
        bw.add_dL_dopacity = g_op, bw.add_dL_dcolors = g_col, bw.add_dL_dmeans3D = g_m3, bw.add_dL_dcov3D = g_cov, bw.add_dL_dsh = g_sh;
        bw.add_dL_dscales = g_sc, bw.add_dL_drotations = g_rot, bw.wait_before_per_gaussian = ready;
        if (hgs_rasterize_backward(&bw, stream) < 0) return fprintf(stderr, "backward (add): %s\n", hgs_last_error()), 3;
        CHECK(hipStreamSynchronize(stream));
        bw.add_dL_dsh = nullptr;   // an incomplete set of add_* pointers is an argument error
        if (M > 0 && hgs_rasterize_backward(&bw, stream) != HGS_ERR_INVALID_ARGUMENT) return fprintf(stderr, "an incomplete add_* set was accepted\n"), 3;
    }

    FILE* o = fopen(argv[2], "wb");
    if (!o) return perror(argv[2]), 1;
    fwrite(&N, 8, 1, o);
    write_dev(o, color, 3 * (size_t)H * W);
    write_dev(o, radii, P);
    write_dev(o, bw.dL_dmeans3D, 3 * (size_t)P);
    write_dev(o, bw.dL_dmeans2D, 3 * (size_t)P);
    write_dev(o, bw.dL_dopacity, P);
    write_dev(o, bw.dL_dsh, 3 * (size_t)M * P);
    if (cov_mode) write_dev(o, bw.dL_dcov3D, 6 * (size_t)P);
    else {
        write_dev(o, bw.dL_dscales, 3 * (size_t)P);
        write_dev(o, bw.dL_drotations, 4 * (size_t)P);
    }
    fclose(o);
    for (void* p : g_scratch) (void)hipFree(p);
    printf("ok N=%lld binning_capacity=%lld\n", (long long)N, (long long)bw.state.binning_capacity);
    return 0;
}
