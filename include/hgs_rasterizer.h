/*
 * hgs_rasterizer.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the one native dependency on the hot path of
 * apple/ml-hugs: the module `diff_gaussian_rasterization`, imported at
 *   /root/reference/hugs/renderer/gs_renderer.py:11-14
 * and used only at
 *   /root/reference/hugs/renderer/gs_renderer.py:126-152.
 * Upstream exposes three C++/pybind entry points from its `_C` extension
 * (rasterize_gaussians, rasterize_gaussians_backward, mark_visible); the
 * three hgs_* entry points below replace them one for one with a plain C ABI:
 * raw device pointers + sizes, no torch types.  The Python side
 * (ml-hugs_amd/diff_gaussian_rasterization) binds them with ctypes.
 *
 * Ownership: every pointer is a device pointer borrowed for the duration of the call.
 * The library allocates nothing that outlives a call: per-call scratch (geometry, binning and
 * image state) is obtained through the caller's allocation callback so that it lives in
 * caller-owned memory (torch uint8 tensors kept by the autograd ctx until backward).
 * All work is enqueued on `stream` (a hipStream_t).  N, the number of (tile, Gaussian) pairs that sizes the binning
 * buffer, reaches the host through ONE 64-bit system-scope store of the tile-scan kernel into a pinned slot the host
 * polls (no copy, no event): forward waits for it before enqueueing the rest of the frame, or, when the caller passes
 * binning_capacity_hint, after it (the wait then overlaps the GPU's work).
 * All floating point is fp32, contiguous.
 *
 * Below the three rasterizer entry points (and their helpers) the header declares the rows either side of the rasterizer
 * that the same library carries, each replacing a named function of the reference: densification statistics (f-1), K nearest
 * template vertices / SMPL LBS blends (f-2), learned-LBS skinning (f-2), distCUDA2 (f-4), the photometric loss l1 + SSIM
 * (f-5), SceneGS.forward (f-6), rotation_6d_to_matrix / matrix_to_quaternion (f-7).  Same conventions: device pointers,
 * sizes, a stream, a negative HGS_ERR_* on failure with hgs_last_error() for the text.
 */
#ifndef HGS_RASTERIZER_H
#define HGS_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGS_ABI_VERSION 11

/* scratch buffer ids passed to the allocation callback */
enum { HGS_BUF_GEOM = 0, HGS_BUF_BINNING = 1, HGS_BUF_IMAGE = 2, HGS_BUF_CKPT = 3, HGS_NUM_BUFS = 4 };

/* error codes (negative return values) */
enum {
    HGS_OK = 0,
    HGS_ERR_INVALID_ARGUMENT = -1,
    HGS_ERR_ALLOC = -2,
    HGS_ERR_HIP = -3,
    HGS_ERR_NO_DEVICE = -4,
    HGS_PENDING = -5,        /* hgs_forward_poll: the frame's tile scan has not run yet */
    HGS_ERR_OVERFLOW = -6,   /* hgs_forward_poll: a deferred frame needed more binning entries than it was given; forward /
                                poll: the frame has 2^32 - 16 or more (tile, Gaussian) pairs -- more than list positions can address */
    HGS_ERR_EXPIRED = -7     /* hgs_forward_poll: the frame's result slot has been handed to a later frame (more than 1 024 forwards
                                were issued before this deferred frame was polled): its N is lost, run the frame again */
};

/* Returns a device pointer to at least `bytes` bytes, 256-byte aligned, or NULL. */
typedef void *(*hgs_alloc_fn)(void *ctx, int buffer_id, size_t bytes);

/* Mirrors GaussianRasterizationSettings as filled at gs_renderer.py:126-139
 * (image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
 * sh_degree, campos, prefiltered, debug). bg/viewmatrix/projmatrix/campos are DEVICE pointers
 * (they are CUDA tensors in the reference's settings tuple). Matrices: row-vector convention,
 * flat [4*r + c]. */
typedef struct hgs_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float *bg;         /* [3] */
    float scale_modifier;
    const float *viewmatrix; /* [16] */
    const float *projmatrix; /* [16] */
    int32_t sh_degree;       /* active degree D, 0..3 */
    const float *campos;     /* [3] */
    int32_t prefiltered;
    int32_t debug;           /* !=0: synchronise + check after every stage */
} hgs_settings;

/* An optional SECOND set of Gaussians rendered together with the first: the reference's joint human + scene render
 * concatenates five tensors per training step (/root/reference/hugs/renderer/gs_renderer.py:33-37: human first, scene
 * second) and autograd splits their gradients again on the way back; with a second segment the two models' tensors are
 * read -- and their gradients written -- in place.  Gaussian index = P + i for the i-th Gaussian of the segment, so every
 * index-dependent result (sorted list, radii, dL/dmeans2D) is exactly what the concatenated call produces.  Same kinds
 * of inputs as the first set (SHs or precomputed colours; scales + rotations or precomputed covariances); M may differ. */
typedef struct hgs_segment {
    int32_t P;                   /* 0: no second segment */
    int32_t M;
    const float *means3D;        /* [P,3] */
    const float *shs;            /* [P,M,3] or NULL */
    const float *colors_precomp; /* [P,3] or NULL */
    const float *opacities;      /* [P] */
    const float *scales;         /* [P,3] or NULL */
    const float *rotations;      /* [P,4] or NULL */
    const float *cov3D_precomp;  /* [P,6] or NULL */
} hgs_segment;

/* Inputs/outputs of the forward pass == kwargs of GaussianRasterizer.forward at
 * gs_renderer.py:144-152 plus the two outputs (rendered_image, radii). */
typedef struct hgs_forward_args {
    hgs_settings s;
    int32_t P;                   /* number of Gaussians */
    int32_t M;                   /* SH coefficients stored per Gaussian (shs is [P,M,3]); 0 if shs NULL */
    const float *means3D;        /* [P,3] */
    const float *shs;            /* [P,M,3] or NULL */
    const float *colors_precomp; /* [P,3] or NULL (exactly one of shs / colors_precomp) */
    const float *opacities;      /* [P] */
    const float *scales;         /* [P,3] or NULL */
    const float *rotations;      /* [P,4] (w,x,y,z), not normalised, or NULL */
    const float *cov3D_precomp;  /* [P,6] or NULL (exactly one of (scales,rotations) / cov3D_precomp) */
    /* Inputs the library rejects PER GAUSSIAN (radius 0, no list entries, zero gradients) beyond the published culls (view-space
     * z <= 0.2; det == 0; an empty tile rectangle): a Gaussian whose PROJECTED 2-D covariance (a, b, c) -- after the 0.3 low-pass --
     * is not positive definite, i.e. not (a c - b^2 > 0 and a > 0).  R S^2 R^T + 0.3 I always is; only a cov3D_precomp that is not
     * positive semi-definite gets there (the reference never passes one: gs_renderer.py:144-152 hands over scales + rotations).
     * The published algorithm rasterizes such a Gaussian and blends it wherever its exponent happens to be
     * <= 0 -- a hyperbolic region of the image; the blend kernels here evaluate the exponent through the conic's Cholesky factors
     * (hgs_common.h), which exist only for a positive-definite conic.  oracle/hgs_oracle.c applies the same rule behind
     * oracle_set_cull_non_pd(1); tests/test_gpu_corner_semantics.py holds the library to it and shows where the two semantics part.
     * Non-finite centres, radii or covariances are culled as well (NaN compares false throughout). */
    float *out_color;            /* [3,H,W], written for every pixel when P > 0 */
    int32_t *radii;              /* [P (+ seg2.P)] */
    /* Optional guess (entries) of N, the number of (tile, Gaussian) pairs -- e.g. last frame's N plus a margin; 0 =
     * none.  With a guess the binning buffer is allocated and the whole frame enqueued before N is known, so the GPU
     * never waits for the host; if the frame needs more than the guess, binning + blending are enqueued again with
     * the exact size (results are identical either way). */
    int64_t binning_capacity_hint;
    /* Optional [P,12] floats: the `grad_accum` the caller is going to hand to hgs_rasterize_backward for this frame.
     * When non-NULL forward zeroes it (inside its first kernel, for free), so the caller need not. */
    float *grad_accum_to_zero;
    /* !=0: out_color = clamp(colour, 0, 1), and backward passes dL/dout_color only where the unclamped value was inside
     * [0, 1] -- exactly `torch.clamp(rendered_image, 0.0, 1.0)` of /root/reference/hugs/renderer/gs_renderer.py:153 and
     * its autograd backward, without the five elementwise passes over the image they cost. */
    int32_t clamp_output;
    /* !=0: the caller expects no long tile -- a list the scan kernel calls LONG: see the generated table "path selection" at the end of this header -- (its previous frame of this shape had none, see
     * hgs_forward_state.has_long_tiles): the long-tile sort kernel is then not launched with the optimistically enqueued
     * frame.  A wrong guess costs that launch plus a second forward blend; results are identical either way. */
    int32_t expect_no_long_tiles;
    /* !=0 (needs binning_capacity_hint > 0): DEFERRED frame -- everything is enqueued for a binning buffer of
     * binning_capacity_hint entries and the call returns 0 at once, without waiting for N (the GPU never waits for the
     * host anyway; this takes the host's one wait per frame away too: forward-only frame loops pipeline freely).  The
     * caller MUST later call hgs_forward_poll(state): it yields N, or HGS_ERR_OVERFLOW when the frame needed more than
     * the hint -- its kernels then did nothing and out_color / radii are NOT valid until the frame is run again.  Use a
     * generous hint (the scratch is the caller's: e.g. a persistent arena per stream). */
    int32_t defer_n;
    /* !=0: the caller is going to run hgs_rasterize_backward on this frame and offers the forward blend a fourth scratch
     * buffer (HGS_BUF_CKPT, hgs_ckpt_bytes(capacity, H, W)) for per-pixel CHECKPOINTS: on a sparse frame (few non-empty
     * tiles with deep lists: a human-only render) the forward then stores (T, colour prefix) of every pixel every 32
     * positions of its quad's list, and backward splits every list into 32-entry segments that run as independent waves
     * instead of one chain of dependent entries per quad.  On a dense frame only the DEEP tiles (512 entries and more: a
     * person in front of a scene) do that, the others go through the one-wave-per-tile backward as always; the library
     * uses the buffer on a dense frame only when the shape's last frame on this stream had a tile list of more than 2048
     * entries (its own record; without one it follows the flag), so a caller that sets this flag for every frame pays
     * only where it helps.  Results are the same up to fp32 summation
     * order either way; 0 keeps the one-wave backward (a failed HGS_BUF_CKPT allocation is an error). */
    int32_t backward_checkpoints;
    /* Optional caller-provided scratch (e.g. persistent arenas for frames that need no backward): buffer k
     * (HGS_BUF_GEOM / HGS_BUF_BINNING / HGS_BUF_IMAGE / HGS_BUF_CKPT) is used when scratch[k] != NULL and
     * scratch_bytes[k] suffices, otherwise the allocation callback is asked as usual.  256-byte aligned device pointers. */
    void *scratch[4];
    size_t scratch_bytes[4];
    hgs_segment seg2;            /* optional second set of Gaussians (all zero: none); scratch / hint sizes count P + seg2.P */
    /* Optional [P (+ seg2.P)] bytes: visible[i] = (radii[i] > 0) -- the `visibility_filter` the reference's render() derives with
     * a separate elementwise kernel (/root/reference/hugs/renderer/gs_renderer.py:159), written by the kernel that writes radii. */
    uint8_t *visible;
    /* Optional guess (checkpoint SLOTS; 0 = none) of what the frame's checkpoints need -- e.g. the previous frame's
     * hgs_forward_state.ckpt_slots_used plus a margin -- for a frame that sets backward_checkpoints and binning_capacity_hint: the
     * HGS_BUF_CKPT buffer is then hgs_ckpt_bytes_for_slots(guess) instead of hgs_ckpt_bytes(capacity): a dense frame leaves
     * checkpoints on its deep tiles only and uses a fraction of the 128 bytes per list entry the full layout holds.  A frame that
     * needs more is detected on the device and run again, exactly sized, like one that overflows its binning buffer (results are
     * identical either way; a deferred frame reports HGS_ERR_OVERFLOW from hgs_forward_poll instead). */
    int64_t ckpt_slots_hint;
    /* Optional: called once, on the calling thread, after the frame's kernels have been enqueued and right BEFORE the call waits for
     * N (not on a deferred frame, which does not wait).  Host work that does not need N -- a binding's autograd bookkeeping -- then
     * runs while the GPU works towards N instead of between N's arrival and the caller's next launch (the backward's, which the GPU
     * reaches ~45 us later on a human-only frame).  Must not call back into the library on this stream. */
    void (*before_wait)(void *ctx);
    void *before_wait_ctx;
} hgs_forward_args;

/* Scratch handed back by forward and required by backward. */
typedef struct hgs_forward_state {
    void *geom;    size_t geom_bytes;
    void *binning; size_t binning_bytes;
    void *image;   size_t image_bytes;
    void *ckpt;    size_t ckpt_bytes;   /* NULL / 0 unless backward_checkpoints was set */
    int64_t num_rendered;     /* N = sum of tiles touched */
    int64_t binning_capacity; /* entries the binning buffer was laid out for (>= N) */
    int32_t sparse_frame;     /* !=0: the scan's SPARSE kind (few non-empty tiles, or deep lists: binning.hip frame_is_sparse); backward gives every 8x8 quad (with checkpoints: every 32-entry segment of its list) its own wave */
    int32_t has_long_tiles;   /* !=0: some tile list is long by the scan kernel's rule (see expect_no_long_tiles; feeds the next frame's expect_no_long_tiles) */
    uint64_t n_token;         /* where hgs_forward_poll finds this frame's N (deferred frames: num_rendered = -1 until polled) */
    int64_t ckpt_slots;       /* slots the checkpoint buffer was laid out for (0: none) */
    int64_t ckpt_slots_used;  /* slots the frame's checkpoints need: (N >> 5) + tiles on a sparse frame, the deep tiles' packed count on a
                                 dense one (0 when it left none); feeds the next frame's ckpt_slots_hint.  -1: a sparse frame that
                                 leaves none BY RULE (7 168 non-empty tiles and more, no heavy tail: its backward runs one wave per quad
                                 without them) -- the caller need not offer this shape's next frame a buffer */
} hgs_forward_state;

/* Replaces _C.rasterize_gaussians. Returns N >= 0, or a negative HGS_ERR_* code. */
int64_t hgs_rasterize_forward(const hgs_forward_args *args, hgs_alloc_fn alloc, void *alloc_ctx,
                              hgs_forward_state *state_out, void *stream);

/* Replaces _C.rasterize_gaussians_backward.  `grad_accum` ([P,12] floats per Gaussian: nine raw sums over the pixels
 * -- u dx, u dy, u dx^2, u dx dy, u dy^2, u (u = G dL/dalpha), dL/dr, dL/dg, dL/db -- and 3 pad) is scratch that must
 * be ZERO on entry (zeroed by the caller, or by forward through fwd.grad_accum_to_zero): the blend-backward kernel
 * accumulates into it with float atomics.  Every dL_* output is fully overwritten (zeros for culled Gaussians and
 * for SH coefficients above the active degree); no pre-zeroing needed. */
typedef struct hgs_backward_args {
    hgs_forward_args fwd;       /* same inputs as forward (out_color unused; radii = forward's output) */
    hgs_forward_state state;    /* as returned by forward */
    const float *dL_dout_color; /* [3,H,W] */
    float *grad_accum;          /* [P,12] scratch, zero on entry */
    float *dL_dmeans2D;         /* [P,3]  (x,y NDC-scaled; z = 0) */
    float *dL_dopacity;         /* [P] */
    float *dL_dcolors;          /* [P,3]  dL/d(colors_precomp) (= dL/d(SH->RGB result) when shs are used) */
    float *dL_dmeans3D;         /* [P,3] */
    float *dL_dcov3D;           /* [P,6] */
    float *dL_dsh;              /* [P,M,3] or NULL */
    float *dL_dscales;          /* [P,3] */
    float *dL_drotations;       /* [P,4] */
    /* With fwd.seg2: grad_accum and dL_dmeans2D cover all P + seg2.P Gaussians (one viewspace tensor, as the reference's
     * joint render has); the seven per-input gradients of the second segment go to its own buffers: */
    float *seg2_dL_dopacity;    /* [P2] */
    float *seg2_dL_dcolors;     /* [P2,3] */
    float *seg2_dL_dmeans3D;    /* [P2,3] */
    float *seg2_dL_dcov3D;      /* [P2,6] */
    float *seg2_dL_dsh;         /* [P2,M2,3] or NULL */
    float *seg2_dL_dscales;     /* [P2,3] */
    float *seg2_dL_drotations;  /* [P2,4] */
    uint32_t flags;             /* HGS_BWD_* bits */
    uint32_t reserved;
    /* Optional (all NULL: none): gradients of the FIRST set's Gaussians left by ANOTHER render's backward of the same step -- the
     * separate human-only render next to the joint one (/root/reference/hugs/renderer/gs_renderer.py:56,69: both differentiate the
     * same human tensors).  The per-Gaussian kernel adds them to its own before storing, so dL_dopacity .. dL_drotations of the
     * first set come out as the SUM of the two renders' gradients -- what autograd otherwise forms with one elementwise kernel per
     * tensor.  Same shapes as the first set's outputs; the other render ran with the same sh_degree.  May not alias the outputs. */
    const float *add_dL_dopacity;
    const float *add_dL_dcolors;
    const float *add_dL_dmeans3D;
    const float *add_dL_dcov3D;
    const float *add_dL_dsh;
    const float *add_dL_dscales;
    const float *add_dL_drotations;
    /* Optional hipEvent_t: `stream` waits for it BETWEEN the blend backward and the per-Gaussian kernel (hipStreamWaitEvent) --
     * the other render's backward, running on another stream, records it when the add_* buffers are complete; the blend
     * backward, the bulk of this call, does not wait. */
    void *wait_before_per_gaussian;
} hgs_backward_args;

/* hgs_backward_args.flags.  dL/dscales is, by default, the true derivative -- it carries settings.scale_modifier, the factor
 * between the stored scale and the one the covariance is built from.  The published CUDA kernel omits that factor; with this
 * bit the library does too.  (Every gradient-enabled call of the reference passes scale_modifier 1.0, where the two agree:
 * /root/reference/hugs/renderer/gs_renderer.py:26,103.) */
#define HGS_BWD_UPSTREAM_SCALE_GRAD 1u

int32_t hgs_rasterize_backward(const hgs_backward_args *args, void *stream);

/* For a frame enqueued with defer_n: N (and state->num_rendered / sparse_frame / has_long_tiles filled in), HGS_PENDING
 * when block == 0 and the frame's tile scan has not run yet, HGS_ERR_OVERFLOW (see defer_n), or another error code.
 * hgs_rasterize_backward refuses a state this call has not resolved (num_rendered < 0). */
int64_t hgs_forward_poll(hgs_forward_state *state, int32_t block, void *stream);

/* Replaces _C.mark_visible: present[i] = (z_view(means3D[i]) > 0.2). */
int32_t hgs_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *present,
                         void *stream);

/* SURVEY.md 8f row f-1 -- fused densification statistics, the trainer-side consumer of the rasterizer's outputs
 * (/root/reference/hugs/trainer/gs_trainer.py:406-411,429-435 and hugs/models/scene.py:460-462): for the first n
 * Gaussians with visibility_filter[i] != 0, in place:
 *   max_radii2D[i] = max(max_radii2D[i], radii[i]);  xyz_gradient_accum[i] += ||viewspace_grad[i, 0:2]||;  denom[i] += 1
 * viewspace_grad is the [>= n, 3] gradient the rasterizer left in viewspace_points.grad. */
int32_t hgs_densification_stats(int32_t n, const float *viewspace_grad, const int32_t *radii,
                                const uint8_t *visibility_filter, float *max_radii2D, float *xyz_gradient_accum,
                                float *denom, void *stream);

/* SURVEY.md 8f row f-2 -- the K nearest template vertices of every query point: replaces pytorch3d.ops.knn_points for
 * batch size 1 as called at /root/reference/hugs/models/hugs_wo_trimlp.py:60,99 (points [n,3], template_points [m,3],
 * 1 <= K <= 8, K <= m).  dists [n,K] = squared L2 distances in ascending order, idx [n,K] int64; equal distances keep
 * the lower template index first.  Pointers need float alignment only. */
int32_t hgs_knn_points(int32_t n, const float *points, int32_t m, const float *template_points, int32_t K,
                       float *dists, int64_t *idx, void *stream);
/* The searches above and below scan the whole template for every point (n x m distances).  Given a workspace of
 * hgs_knn_workspace(n, m) bytes (16-byte aligned, contents don't matter; 0 = the grid would not pay: a template under 512
 * vertices or fewer than 4 points per vertex, the *_ws forms then scan as well) the *_ws forms first put the template on a
 * uniform grid, sort the points by the cell they fall into, and let every 64 neighbouring points test only the vertices of
 * the cells around them; a point whose K-th neighbour is not certainly inside that box scans the whole template in a second
 * pass.  The same neighbours in the same order, exact ties included, in a fraction of the distances (110 000 points x 6 890
 * SMPL vertices: 0.38 -> 0.17 ms). */
size_t hgs_knn_workspace(int32_t n, int32_t m);
int32_t hgs_knn_points_ws(int32_t n, const float *points, int32_t m, const float *template_points, int32_t K,
                          float *dists, int64_t *idx, void *workspace, void *stream);

/* Replaces smpl_lbsweight_top_k (hugs_wo_trimlp.py:88-119; called on every training step at hugs_trimlp.py:318,480)
 * for batch size 1, search and blending fused: lbs_weights [m,J] (J = 24 for SMPL), out_dist [n] (xyz_dist),
 * out_weights [n,J] (the K-neighbour blend of the template's LBS weights, confidence-gated as upstream). */
int32_t hgs_smpl_lbsweight_top_k(int32_t n, const float *points, int32_t m, const float *template_points,
                                 const float *lbs_weights, int32_t J, int32_t K, float *out_dist, float *out_weights,
                                 void *stream);
int32_t hgs_smpl_lbsweight_top_k_ws(int32_t n, const float *points, int32_t m, const float *template_points,
                                    const float *lbs_weights, int32_t J, int32_t K, float *out_dist, float *out_weights,
                                    void *workspace, void *stream);

/* Replaces smpl_lbsmap_top_k (hugs_wo_trimlp.py:47-85, the model variant without the triplane) for batch size 1, search and
 * blending fused: out_transform [n,16] = sum_k wgt_k verts_transform[idx_k] ([m,16], row-major 4x4, 16-byte aligned),
 * out_info [n,C] likewise from addition_info [m,C] (both NULL: none), out_dist [n]; out_idx [n,K] (int32) / out_wgt [n,K]
 * are the neighbours and their normalised, confidence-gated weights, which the backward takes: gradients with respect to
 * verts_transform / addition_info ([m,16] / [m,C], ZERO on entry: float atomics add into them), as the reference
 * differentiates (the search and the weights are constants there too). */
int32_t hgs_smpl_lbsmap_top_k(int32_t n, const float *points, int32_t m, const float *template_points,
                              const float *lbs_weights, int32_t J, int32_t K, const float *verts_transform,
                              const float *addition_info, int32_t C, float *out_dist, float *out_transform,
                              float *out_info, int32_t *out_idx, float *out_wgt, void *workspace /* hgs_knn_workspace(n, m) or NULL */,
                              void *stream);
int32_t hgs_smpl_lbsmap_top_k_backward(int32_t n, int32_t K, const int32_t *idx, const float *wgt,
                                       const float *dL_dtransform, const float *dL_dinfo, int32_t C,
                                       float *dL_dverts_transform, float *dL_daddition_info, void *stream);

/* SURVEY.md 8f row f-2, second half -- the skinning step of the human model's learned LBS, fused: replaces the matmul /
 * cat / batched-matmul / slice statements of lbs_extra (/root/reference/hugs/models/modules/lbs.py:60-73, called every
 * training step at hugs/models/hugs_trimlp.py:477-489) and the rotation product of hugs_trimlp.py:517, for one batch element:
 *   T[i] = sum_j weights[i,j] A[j]   ([n,16], row-major 4x4)     verts[i] = (T[i] [v[i], 1])[:3]
 *   rot_out[i] = T[i][:3,:3] rotmat[i]                            (rotmat / rot_out may both be NULL)
 * A [J,16], 1 <= J <= 32; weights [n,J]; v [n,3] (= v_posed); rotmat [n,9].  T must be 16-byte aligned. */
int32_t hgs_lbs_skin_forward(int32_t n, int32_t J, const float *A, const float *weights, const float *v,
                             const float *rotmat, float *T, float *verts, float *rot_out, void *stream);
/* Its backward: gradients w.r.t. A [J,16], weights [n,J], v [n,3] and rotmat [n,9] given dL/dverts, dL/dT, dL/drot_out
 * (each may be NULL = zero).  T is forward's output.  `workspace`: hgs_lbs_skin_backward_workspace(n, J) bytes, 16-byte
 * aligned.  dL/dA is a contraction over all n Gaussians: it runs on the matrix cores and is reduced in a fixed order (no
 * float atomics, bit-reproducible). */
size_t hgs_lbs_skin_backward_workspace(int32_t n, int32_t J);
int32_t hgs_lbs_skin_backward(int32_t n, int32_t J, const float *A, const float *weights, const float *v,
                              const float *rotmat, const float *T, const float *dL_dverts, const float *dL_dT,
                              const float *dL_drot, float *dL_dA, float *dL_dweights, float *dL_dv, float *dL_drotmat,
                              void *workspace, void *stream);

/* SURVEY.md 8f row f-4 -- replaces simple_knn._C.distCUDA2 (/root/reference/hugs/models/scene.py:20,181): mean_dist2[i]
 * = mean of the squared distances from points[i] to its three nearest OTHER points of the same cloud ([n,3], n >= 4).
 * Exact (brute force, O(n^2)), fp32. */
int32_t hgs_dist_cuda2(int32_t n, const float *points, float *mean_dist2, void *stream);
/* The same result (bit for bit) in O(n) for large clouds: with a workspace of hgs_dist_cuda2_workspace(n) bytes (16-byte
 * aligned; 0 = this n takes the brute-force scan anyway) the points are counting-sorted into a uniform grid and every point
 * searches the shells of cells around its own until its third-best distance is closed. */
size_t hgs_dist_cuda2_workspace(int32_t n);
int32_t hgs_dist_cuda2_ws(int32_t n, const float *points, float *mean_dist2, void *workspace, void *stream);

/* Row f-5 -- the photometric loss that consumes the rendered image on every training step, fused: replaces l1_loss and ssim
 * (/root/reference/hugs/losses/utils.py:54-58,65-108; called at hugs/losses/loss.py:88-107 and again for the human-only render
 * at :128-137).  img1 (the render, differentiable) and img2 (the target) are [C,H,W]; the window is the reference's 11x11
 * Gaussian (sigma 1.5, zero padding, one group per channel), C1 = 0.01^2, C2 = 0.03^2.
 *   out[0] = mean of the SSIM map, out[1] = mean |img1 - img2|, out[2] = sum |img1 - img2| (l1_loss with a mask divides the
 *   sum by mask.sum()).
 * maps: [3,C,H,W] floats kept for the backward, or NULL for a forward-only evaluation.  workspace: hgs_ssim_l1_workspace(C,H,W)
 * bytes, 8-byte aligned (per-workgroup partial sums; the totals are formed in a fixed order in double: bit-reproducible). */
size_t hgs_ssim_l1_workspace(int32_t C, int32_t H, int32_t W);
int32_t hgs_ssim_l1_forward(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *maps,
                            void *workspace, float *out, void *stream);
/* dL/dimg1 [C,H,W] = g_ssim_mean[0] * d(out[0])/dimg1 + g_l1_sum[0] * d(out[2])/dimg1; both factors are DEVICE scalars (the
 * autograd gradients of the two outputs: no host round trip), either may be NULL (= 0).  maps: forward's, needed for the
 * SSIM term. */
int32_t hgs_ssim_l1_backward(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, const float *maps,
                             const float *g_ssim_mean, const float *g_l1_sum, float *dL_dimg1, void *stream);

/* Row f-6 -- the statements that produce the rasterizer's inputs on every training step, fused: replaces SceneGS.forward
 * (/root/reference/hugs/models/scene.py:147-160): scales = exp(scaling) [P,3], rotq = normalize(rotation) [P,4] (x / max(|x|,
 * 1e-12), NOT unit-length input), opacities = sigmoid(opacity) [P,1], shs = cat(features_dc [P,1,3], features_rest [P,M-1,3])
 * -> [P,M,3].  rotation / rotq must be 16-byte aligned. */
int32_t hgs_scene_forward(int32_t P, int32_t M, const float *scaling, const float *rotation, const float *opacity,
                          const float *features_dc, const float *features_rest, float *scales, float *rotq,
                          float *opacities, float *shs, void *stream);
/* Its backward.  Any of the four incoming gradients may be NULL (= zero: the matching outputs are then not written). */
int32_t hgs_scene_backward(int32_t P, int32_t M, const float *rotation, const float *scales, const float *opacities,
                           const float *dL_dscales, const float *dL_drotq, const float *dL_dopacities, const float *dL_dshs,
                           float *dL_dscaling, float *dL_drotation, float *dL_dopacity, float *dL_dfeatures_dc,
                           float *dL_dfeatures_rest, void *stream);

/* Row f-7 -- the rotation conversions on the human model's forward path that produce the rasterizer's `rotations` argument
 * (/root/reference/hugs/models/hugs_trimlp.py:418-419,518), one thread per rotation, no host synchronisation:
 *   hgs_rotation_6d_to_matrix: d6 [n,6] -> matrix [n,9] (rows b1, b2, b3), /root/reference/hugs/utils/rotations.py:552-573
 *   hgs_matrix_to_quaternion:  matrix [n,9] row-major -> quat [n,4] (w,x,y,z), rotations.py:94-156 -- the candidate with the
 *     largest denominator, floor 0.1, exactly as stated there; quat / dL_dquat must be 16-byte aligned.
 * The backward forms take the forward's INPUT (everything is recomputed from it) and the gradient of its output. */
int32_t hgs_rotation_6d_to_matrix(int32_t n, const float *d6, float *matrix, void *stream);
int32_t hgs_rotation_6d_to_matrix_backward(int32_t n, const float *d6, const float *dL_dmatrix, float *dL_dd6, void *stream);
int32_t hgs_matrix_to_quaternion(int32_t n, const float *matrix, float *quat, void *stream);
int32_t hgs_matrix_to_quaternion_backward(int32_t n, const float *matrix, const float *dL_dquat, float *dL_dmatrix, void *stream);

/* Message for the last negative return value on the calling thread. */
const char *hgs_last_error(void);

int32_t hgs_abi_version(void);

/* Scratch sizes (so a caller may pre-allocate) */
size_t hgs_geom_bytes(int32_t P, int32_t image_height, int32_t image_width);
size_t hgs_image_bytes(int32_t image_height, int32_t image_width);
size_t hgs_binning_bytes(int64_t num_rendered, int32_t image_height, int32_t image_width);
size_t hgs_ckpt_bytes(int64_t num_rendered, int32_t image_height, int32_t image_width);
size_t hgs_ckpt_bytes_for_slots(int64_t slots);   /* (hgs_forward_args.ckpt_slots_hint) */

/* Per-stage device timing with HIP events recorded on the launch stream (the
 * reference has no profiling hooks; this is the build's own).  `stage_mask` has bit k set to time
 * stage k; 0 disables (default).  hgs_profile_read synchronises the recorded events of that stage,
 * adds them up since the last reset and returns the number of timed launches through *launches. */
enum {
    HGS_STAGE_PREPROCESS = 0,    /* projection + SH + the first step of the binning (ranks inside screen cells on large
                                    frames, per-tile pair counts on small ones) */
    HGS_STAGE_SCAN = 1,          /* large frames: cell scatter + per-group pair counts; then the tile scan -> ranges, N */
    HGS_STAGE_EMIT_KEYS = 2,
    HGS_STAGE_SORT = 3,          /* per-tile sort, fused with the forward blend unless HGS_FUSED_SORT_BLEND=0 */
    HGS_STAGE_BLEND_FORWARD = 4, /* stand-alone forward blend (unfused runs, repaired long tiles) */
    HGS_STAGE_BLEND_BACKWARD = 5,
    HGS_STAGE_PREPROCESS_BACKWARD = 6, HGS_NUM_STAGES = 7
};
void hgs_profile_enable(uint32_t stage_mask);
/* Time only every n-th launch of an enabled stage (default 1: every launch).  An event pair costs a few microseconds of
 * GPU time around the kernel it brackets: a throughput measurement that also wants a live per-kernel figure samples. */
void hgs_profile_set_sampling(uint32_t every_nth);
int32_t hgs_profile_read(int32_t stage, double *total_ms, int64_t *launches);
void hgs_profile_reset(void);
const char *hgs_stage_name(int32_t stage);

/* Measurement aid: one float4-per-thread device-to-device copy of `bytes` (a multiple of 16) on `stream` -- bench.py times
 * it with events for `roofline.peak_measured`, the practical HBM ceiling of the GPU it runs on. */
int32_t hgs_copy_bandwidth(void *dst, const void *src, size_t bytes, void *stream);

/* Test/debug introspection: byte offsets of the named sub-arrays inside the scratch buffers.
 * Names: geom: "splats" (64-byte records), "tiles_touched"; binning: "list" (the sorted list, one u64 per entry:
 * (1-based position inside the tile << 32) | quad coverage mask << 28 | Gaussian index);
 * image: "final_T", "n_contrib" (low 28 bits: position of the last contributing entry; bits 29..31: which colour channels
 * pass dL/dout_color, all set unless clamp_output clipped them), "ranges". Returns (size_t)-1 for an unknown name. */
size_t hgs_scratch_offset(const char *name, int32_t P, int64_t num_rendered, int32_t image_height,
                          int32_t image_width);

/* Test/debug introspection of the library's own (host-side) state: "tile_counter_entries" (per-stream counter arrays it
 * currently keeps), "tile_counter_max_entries" (the bound beyond which idle streams' arrays are dropped), "slot_ring"
 * (result slots: forwards after which an unpolled deferred frame expires); host-time accounting since the library was loaded:
 * "forward_calls" / "forward_ns" (time inside hgs_rasterize_forward) / "forward_wait_ns" (the part of it spent waiting for N),
 * "backward_calls" / "backward_ns"; "binning_reruns" / "ckpt_reruns" (optimistically enqueued frames that were run again because
 * they needed more binning entries / checkpoint slots than guessed).  -1 for an unknown name. */
int64_t hgs_debug_stat(const char *name);

/* The library reads its A/B switches (HGS_BIN_MODE, HGS_BWD_TWO_LAUNCHES, HGS_DEEP_FORWARD, HGS_LONG_MIN_SPARSE, HGS_LONG_MIN_DENSE, HGS_EMIT_SCAN, HGS_K1_STAGE_SH, HGS_BIG_PER_GROUP)
 * from the environment once, at its first frame; a test or A/B tool that changes them inside one process calls this afterwards. */
void hgs_reload_switches(void);

/* How the library chooses its path for a frame.  Every decision is taken from the frame's own numbers (non-empty tiles, list lengths) or,
 * where the host has to decide before the first kernel, from the record of the shape's last frame -- which only ever sizes launches and
 * buffers or picks between two forms with identical results (gradients: up to fp32 summation order).  Images, radii and lists never depend on
 * history.  The numbers below are read out of the kernels' sources:
 *
 * BEGIN GENERATED: path selection (tools/gen_thresholds.py: do not edit by hand)
 * - frame kind: DENSE (one backward wave per tile) or SPARSE (a wave per 8x8 quad; from the forward's checkpoints: per 32-entry segment):
 *     n = non-empty tiles, E = sum(len^2) / N (the list length a random entry sits in), mean = N / n.  n >= 4 096: dense unless E > 830 (+ up to 680 more below 8 192 tiles, linearly: 830 + 680 at 4 096) AND the depth is the frame's own: E <= 26 / 10 mean (a heavy tail on a covered frame -- a person in front of a scene -- stays dense: its deep tiles take the checkpointed walk) and the frame is not flat (longest list <= 1.25 E, E <= 1 600: dense).  1 536 <= n < 4 096: dense while E <= min(1 200, 0.45 (n - 800)) -- up to 1 600 on a flat frame (longest list <= 1.25 E) -- and E <= 2.5 mean.  n < 1 536: sparse
 *     [DENSE_ALWAYS_TILES, DENSE_ALWAYS_E_MAX, DENSE_ALWAYS_E_RISE, DENSE_ALWAYS_TAIL_X10, DENSE_MIN_TILES, DENSE_E_MAX, DENSE_E_ORIGIN, DENSE_E_FLAT_MAX (binning.hip, frame_is_sparse)]
 * - checkpoints for the depth-segmented backward (when the caller offers a buffer):
 *     sparse frame: every tile -- none when n >= 7 168 and E < 1.6 mean (hgs_forward_state.ckpt_slots_used = -1).  dense frame: its tiles of >= 512 entries, and only when the shape's last frame held a list beyond 2 048 entries (host, from the shape's record)
 *     [CKPT_DEEP_MIN, DEEP_BWD_MIN, CKPT_SEG = 32 entries per segment (hgs_common.h), NO_CKPT_MIN_TILES (binning.hip)]
 * - LONG lists (sorted ahead of the fused kernel by the long tiles' kernels):
 *     sparse frame: beyond 256 entries when mean >= 200 and 16 .. 512 lists are that long; else beyond 1 024 when 16 .. 512 lists are; with more than 512 lists beyond 1 024: beyond 1 024 if the longest list is <= 4 096 (flat), else beyond 2 048.  dense frame: none unless the frame holds a list beyond 2 048; then beyond 768, or beyond 1 024 when more than 832 lists lie beyond 768
 *     [LONG_MIN_SPARSE, DEEP_MEAN_MIN, LONG_MIN_SPARSE_TILES, LONG_ONE_ROUND, LONG_MIN_SPARSE_SHALLOW, LONG_MIN_DENSE, DENSE_LONG_MANY, SORT_CAP_SMALL, SORT_CAP_MID (binning.hip, tile_scan_body)]
 * - long tiles blended split by depth (four waves per quad: the deep workers):
 *     dense frames; sparse frames with mean >= 200 -- except more than 512 long lists none of which is beyond 4 096 entries (flat: one wave per quad), and except sparse frames of >= 1 000 non-empty tiles whose longest list is <= 18 / 10 E (full and even: a person filling the frame)
 *     [n_total[8], DEEP_EVEN_TILES, DEEP_EVEN_L_X10 (binning.hip); HGS_DEEP_FORWARD=0 / HGS_DEEP_MIN override]
 * - per-tile sort inside the fused kernel:
 *     <= 256 entries: bitonic network in registers; <= 1 024: bucket sort in LDS; <= 2 048: bitonic network, eight keys per thread; long tiles' kernel: one workgroup per list of <= 4 096 entries, longer lists split by depth into parts of 3 072 .. 4 096
 *     [SORT_CAP_SMALL, SORT_CAP_MID, PLAN_PART (binning.hip)]
 * - binning groups (host, before the first kernel):
 *     by screen cell (two more launches) when P >= 32 768, tiles >= 4 096, cells <= 2 048 and the shape's last frame covered at least half the tiles; else in storage order.  Per-tile LDS counters: 32-bit up to 22 528 tiles, 16-bit up to 45 056, global atomics beyond.  Splats of more than 256 tiles: groups of their own, 24 each
 *     [bin_mode_for, BIN_LDS_TILES, BIN_LDS16_TILES, BIN_SPREAD_MIN, BIG_PER_GROUP (hgs_common.h)]
 * - tile scan folded into the emit launch (host):
 *     frames enqueued on a capacity guess with <= 8 192 x 2 tiles and <= 1 024 binning groups
 *     [EMIT_SCAN_TILES, EMIT_SCAN_MAX_CHUNKS (binning.hip)]
 * END GENERATED: path selection
 */

#ifdef __cplusplus
}
#endif
#endif /* HGS_RASTERIZER_H */
