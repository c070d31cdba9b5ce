/*
 * gsplat_mi355x.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * Drop-in boundary for the hot path of ashu1069/3D-Gaussian-Splatting-for-Novel-View-Synthesis.
 * The reference has no FFI of its own: its boundary is three Python functions in the
 * `gaussian_splatting` namespace (reference gaussian_splatting/__init__.py:7-21).  This header is
 * what a binding for those three functions links against; INTEGRATION.md shows the ctypes stub.
 *
 *   reference function (file:line)                               replaced by
 *   ------------------------------------------------------------ --------------------------------------
 *   render()                  gaussian_splatting/render.py:62    gsplat_project + gsplat_bin +
 *                                                                gsplat_rasterize_forward (+ the two
 *                                                                *_backward calls for autograd)
 *   build_sigma_from_params() gaussian_splatting/gaussian.py:71  gsplat_build_sigma[_backward], or folded
 *                                                                into gsplat_project (fused inputs)
 *   evaluate_sh()   gaussian_splatting/spherical_harmonics.py:70 gsplat_evaluate_sh[_backward], or folded
 *                                                                into gsplat_project (fused inputs)
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; `stream` is a hipStream_t passed as void*.
 *   - every pointer is a DEVICE pointer to fp32 data in the reference's own tensor layout
 *     (pos[N,3], f_rest[N,45] channel-major, sigma[N,3,3], image[H,W,3] ...), except where a
 *     parameter name ends in `_host`.
 *   - the library never allocates or frees memory the caller can see: inputs, outputs, state kept
 *     for the backward pass and scratch are caller-owned (PyTorch tensors in the Python host).
 *     `*_bytes()` functions give the sizes; the internal carving is private to the library.
 *   - every call is stream-ordered and returns immediately (no host synchronisation inside).
 *   - return value: GSPLAT_OK or a GSPLAT_ERR_* code; gsplat_last_error() gives the text
 *     (thread-local).
 */
#ifndef GSPLAT_MI355X_H
#define GSPLAT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSPLAT_ABI_VERSION 8

/* call status */
#define GSPLAT_OK 0
#define GSPLAT_ERR_BAD_ARG 1     /* null pointer, negative size, unsupported tile size ...          */
#define GSPLAT_ERR_HIP 2         /* a HIP runtime call or kernel launch failed                      */
#define GSPLAT_ERR_WORKSPACE 3   /* caller-provided state / scratch buffer too small                */

/* scene status, from gsplat_classify_counts(): mirrors the reference's empty / error conventions   */
#define GSPLAT_SCENE_OK 0
#define GSPLAT_SCENE_ALL_CULLED 10     /* render.py:109-112,139-142,182-193,206-209 -> zero image   */
#define GSPLAT_SCENE_ALL_OFFSCREEN 11  /* render.py:235-236 -> Exception("All projected points ...") */

/* Per-view scalars: the intrinsics and the 8 keyword arguments of render() (render.py:62-64).      */
typedef struct gsplat_view {
    int32_t H, W;                 /* image size                                                      */
    float fx, fy, cx, cy;         /* pinhole intrinsics (pixels)                                     */
    float near_z, far_z;          /* near=0.01, far=100.0                                            */
    float pix_guard;              /* 32                                                              */
    int32_t tile;                 /* T=16; any T >= 1 (it sets the reference's tile rectangles = the
                                     reported pair count; the image does not depend on it)            */
    float min_conis;              /* 1e-6                                                            */
    float chi_square_clip;        /* 6.25                                                            */
    float alpha_max;              /* 0.99                                                            */
    float alpha_cutoff;           /* 1/128                                                           */
} gsplat_view;

/* Gaussian parameters of one scene.  Exactly one of the two input sets is given:
 *   un-fused (the reference render() signature): color + sigma, the four fused pointers NULL;
 *   fused (build_sigma_from_params + evaluate_sh folded in): scale_raw, q_raw, f_dc, f_rest,
 *          color and sigma NULL.                                                                    */
typedef struct gsplat_gaussians {
    int64_t n;
    const float* pos;           /* [n,3]                                                             */
    const float* opacity_raw;   /* [n]                                                               */
    const float* color;         /* [n,3]   or NULL                                                   */
    const float* sigma;         /* [n,3,3] or NULL                                                   */
    const float* scale_raw;     /* [n,3]   or NULL   (log scale)                                     */
    const float* q_raw;         /* [n,4]   or NULL   (x,y,z,w), un-normalised                        */
    const float* f_dc;          /* [n,3]   or NULL                                                   */
    const float* f_rest;        /* [n,45]  or NULL   channel-major R1..R15,G1..G15,B1..B15           */
} gsplat_gaussians;

/* Gradient outputs, same shapes as the inputs of gsplat_gaussians; unused ones NULL.
 * Every row is written (rows of culled Gaussians get zeros, as in the reference).                    */
typedef struct gsplat_gaussian_grads {
    float* pos;
    float* opacity_raw;
    float* color;
    float* sigma;
    float* scale_raw;
    float* q_raw;
    float* f_dc;
    float* f_rest;
} gsplat_gaussian_grads;

/* Counters produced by gsplat_project (copied to pinned host memory when asked).                    */
typedef struct gsplat_counts {
    int32_t n_survivors;   /* pass the opacity prefilter, frustum cull and finite check              */
    int32_t n_visible;     /* ... and have an on-screen AABB  (V of SURVEY.md)                       */
    int64_t n_pairs;       /* the reference's (tile, Gaussian) pairs, F11  (P of SURVEY.md)          */
    int32_t max_tiles_per_gaussian;   /* of the binned rectangles                                    */
    int32_t reserved;
    int64_t n_binned;      /* (list, Gaussian) pairs actually binned: the ellipse of each Gaussian over
                              16 x 8-pixel lists; sizes the gsplat_bin buffers                        */
} gsplat_counts;

int gsplat_abi_version(void);
const char* gsplat_last_error(void);
int gsplat_classify_counts(const gsplat_counts* counts_host);

/* ---- buffer sizes (bytes) ---------------------------------------------------------------------- */
int64_t gsplat_project_state_bytes(int64_t n, const gsplat_view* v);    /* kept until the backward pass */
int64_t gsplat_project_scratch_bytes(int64_t n);     /* persistent counter block of gsplat_project: see there         */
int64_t gsplat_bin_state_bytes(int64_t pair_capacity, const gsplat_view* v);   /* kept until the backward pass   */
int64_t gsplat_bin_scratch_bytes(int64_t pair_capacity, const gsplat_view* v); /* free after gsplat_bin          */

/* ---- forward ----------------------------------------------------------------------------------- */
/* F1-F8, F10, F13 (+F2, F3 when fused): per-Gaussian projection, culls, EWA covariance, eigen clamp,
 * conic, rectangle and mask of 16 x 8-pixel lists, colour; counts the (list, Gaussian) pairs in total and
 * per coarse bin.  At most 2^26 Gaussians per call.  c2w is the DEVICE [4,4] row-major camera-to-world matrix (no
 * host read -> no synchronisation).
 *   scratch        gsplat_project_scratch_bytes() bytes, 64-byte aligned: a block of counters that must be ZERO when
 *                  the call starts.  Zero it once after allocating it; every call leaves it zeroed again (the last wave
 *                  of the projection kernel adds the counters up and clears them: no clearing or totals kernel).  One
 *                  block per stream; calls sharing a block must be stream-ordered.
 *   counts_host    (nullable) receives the counters: by hipMemcpyAsync on `stream`, or -- flag
 *                  GSPLAT_PROJECT_COUNTS_MAPPED: it is device-accessible pinned host memory (hipHostMalloc) -- stored by
 *                  the kernel itself (one stream operation less).
 *   counts_event   (hipEvent_t, nullable) recorded right behind the counters: a caller that wants exact buffer sizes
 *                  waits for it (not for the stream: the first binning kernel and, without COLOUR_FUSED, the SH colour
 *                  pass are queued BEHIND the event and run during the host's round trip) and reads n_binned.
 *   flags          GSPLAT_PROJECT_COLOUR_FUSED: evaluate the SH colour inside the projection kernel (one pass over the
 *                  inputs: best when the host does NOT wait for the counters -- see gsplat_bin's pair_capacity).
 *                  GSPLAT_PROJECT_SAVE_SH_JACOBIAN (fused inputs; set it when a backward pass will follow): the colour pass
 *                  also leaves, per visible Gaussian, d colour / d logit and d logit / d position (48 bytes) in
 *                  project_state, so that gsplat_project_backward (flag GSPLAT_BACKWARD_SH_JACOBIAN) does not read the
 *                  192 bytes of SH coefficients again.  Ignored for un-fused inputs.
 *                  GSPLAT_PROJECT_COUNTS_LATE (for callers that do NOT wait for the counters in the middle of the forward
 *                  pass): the counters -- device copy, counts_host, counts_event -- are produced by the first binning kernel
 *                  instead of by the projection kernel's last wave; the projection's waves then retire without waiting
 *                  for their stores.  Everything queued after this call sees them as before.                           */
#define GSPLAT_PROJECT_COLOUR_FUSED 1
#define GSPLAT_PROJECT_COUNTS_MAPPED 2
#define GSPLAT_PROJECT_SAVE_SH_JACOBIAN 4
#define GSPLAT_PROJECT_COUNTS_LATE 8
int gsplat_project(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* project_state,
                   void* scratch, int64_t scratch_bytes, gsplat_counts* counts_host, void* counts_event, int32_t flags,
                   void* stream);

/* F9, F11, F12: every Gaussian is appended to the lists of its rectangle (two-level counting sort),
 * the lists get their [start, end) and a longest-first launch order, and every list is sorted; the
 * order inside a list is (camera depth, Gaussian index) ascending.  The rendered image does not
 * depend on the binning granularity, only on that order.
 *   pair_capacity  pairs the bin_state / scratch buffers were sized for (gsplat_bin_*_bytes).  The number of pairs
 *                  really binned is read on the DEVICE (the counters gsplat_project left in project_state), so the
 *                  host may pass the exact n_binned it waited for, or -- without ever synchronising -- a capacity kept
 *                  from earlier frames.  If the frame has more pairs than that, nothing is written out of bounds, the
 *                  frame's image and gradients are garbage, and the caller finds n_binned > pair_capacity in the
 *                  counters whenever it reads them: it then renders the frame again with larger buffers.
 *                  At most 2^32 - 1 pairs; images up to 8192 coarse bins (64 lists each: 8192 x 8192 pixels).          */
int gsplat_bin(int64_t n, int64_t pair_capacity, const gsplat_view* v, const void* project_state, void* bin_state,
               void* scratch, int64_t scratch_bytes, void* stream);

/* F14, F15: per-tile front-to-back compositing.  image[H,W,3] receives clamp(C,0,1); accum[H,W,3]
 * (nullable; required for the backward pass) receives the unclamped C.  grad2d (nullable, [n,16]
 * floats): cleared here for the coming gsplat_rasterize_backward (pass grad2d_zeroed = 1 there), which
 * saves that call a 64-byte-per-Gaussian fill; only worth it when n / lists is small (<= 256).
 * pair_capacity must be the value gsplat_bin was given.  With `accum` (a backward pass will follow) the call
 * also leaves one byte per pair in bin_state -- which 4 x 4-pixel sub-tiles of its list the Gaussian's ellipse
 * reaches, by the exact test -- and gsplat_rasterize_backward composites from those.                    */
int gsplat_rasterize_forward(int64_t n, int64_t pair_capacity, const gsplat_view* v, const void* project_state,
                             void* bin_state, float* image, float* accum, float* grad2d, void* stream);

/* ---- backward ---------------------------------------------------------------------------------- */
/* B1: gradient of the compositing w.r.t. the per-Gaussian 2D quantities.  grad2d is [n,16] floats, private to the
 * library (moments of dL/dq over the pixels for the centre and the conic, then opacity, r, g, b, padding); it is zeroed
 * by this call before accumulation (unless grad2d_zeroed: gsplat_rasterize_forward already cleared it) and consumed
 * by gsplat_project_backward.
 *   det_scratch    NULL: the per-(list, Gaussian) sums are added into grad2d with float atomics (fastest; the order of the
 *                  additions, hence the last bits of the gradients, varies from run to run).  Not NULL
 *                  (gsplat_rasterize_backward_scratch_bytes(n, pair_capacity) bytes): DETERMINISTIC mode -- the sums are
 *                  stored per pair and added per Gaussian in a fixed order: gradients are bitwise reproducible.            */
int64_t gsplat_rasterize_backward_scratch_bytes(int64_t n, int64_t pair_capacity);
int gsplat_rasterize_backward(int64_t n, int64_t pair_capacity, const gsplat_view* v, const void* project_state,
                              const void* bin_state, const float* accum, const float* grad_image,
                              float* grad2d, int32_t grad2d_zeroed, void* det_scratch, int64_t det_scratch_bytes,
                              void* stream);

/* B2 (+B3 when fused): chain the 2D gradients back to the inputs of gsplat_project.
 * Factored form (fused inputs; out->f_dc and out->f_rest NULL): instead of the 48 SH-coefficient
 * gradients per Gaussian, out->color[n,3] (if given) receives the gradient w.r.t. the colour LOGIT (the sigmoid's argument,
 * spherical_harmonics.py:166); gsplat_sh_accumulate turns logit gradients of any number of views into SH gradients.
 *   flags          GSPLAT_BACKWARD_SH_JACOBIAN: project_state was filled by gsplat_project with
 *                  GSPLAT_PROJECT_SAVE_SH_JACOBIAN for the SAME g and c2w (same results up to fp32 rounding, 144 bytes
 *                  less HBM traffic per visible Gaussian).  Without the flag the SH coefficients are read again.         */
#define GSPLAT_BACKWARD_SH_JACOBIAN 1
int gsplat_project_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v,
                            const void* project_state, const float* grad2d, const gsplat_gaussian_grads* out,
                            int32_t flags, void* stream);

/* ---- composite entries: one call per direction ----------------------------------------------------------------------
 * For a host that does not wait for the counters in the middle of the forward pass (a training loop, a frame sequence:
 * the pair buffers are sized from a capacity kept from earlier frames, see gsplat_bin) the forward pass is three library
 * calls and five caller-owned buffers, the backward pass two or three calls; on a Python host every call and every
 * allocation is several microseconds, and the whole step is ~0.5 ms of GPU time.  These two entries queue exactly the same
 * kernels from ONE call each, on ONE caller-owned arena:
 *
 *   frame arena   gsplat_frame_bytes(n, pair_capacity, v, flags) bytes, 256-byte aligned: project_state | bin_state, and with
 *                 GSPLAT_FRAME_BACKWARD also accum [H,W,3] | grad2d [n,16]; kept until the backward pass, private to the library.
 *   gsplat_forward_deferred = gsplat_project(COLOUR_FUSED | COUNTS_LATE | COUNTS_MAPPED if counts_host | SAVE_SH_JACOBIAN if
 *                 GSPLAT_FRAME_BACKWARD and fused inputs) + gsplat_bin + gsplat_rasterize_forward.  `counters` is
 *                 gsplat_project's persistent counter block, `bin_scratch` gsplat_bin's scratch (gsplat_bin_scratch_bytes);
 *                 counts_host (nullable) must be device-mapped pinned memory; counts_event (nullable) is recorded behind the
 *                 counters.  The caller reads counts_host when it likes (n_binned > pair_capacity: the frame is garbage, render
 *                 it again with larger buffers; survivors but nothing visible: the reference's off-screen exception).
 *   gsplat_backward = gsplat_rasterize_backward [+ gsplat_logit_grad if grad_logit] + gsplat_project_backward on the same
 *                 arena.  flags: GSPLAT_BACKWARD_SH_JACOBIAN as for gsplat_project_backward (set it iff the frame was made
 *                 with GSPLAT_FRAME_BACKWARD from fused inputs without GSPLAT_FRAME_NO_SH_JACOBIAN);
 *                 GSPLAT_BACKWARD_PHASE_RASTER / _PROJECT: only that half (a data-parallel host starts exchanging grad_logit
 *                 between the two); GSPLAT_BACKWARD_GRAD2D_DIRTY: a second backward pass through the same frame.           */
#define GSPLAT_FRAME_BACKWARD 1
#define GSPLAT_FRAME_NO_SH_JACOBIAN 2
#define GSPLAT_BACKWARD_PHASE_RASTER 2
#define GSPLAT_BACKWARD_PHASE_PROJECT 4
#define GSPLAT_BACKWARD_GRAD2D_DIRTY 8
/* the gradients are ADDED to what `out` holds (the views of one iteration summed by the projection backward itself; the caller
 * clears or writes the arrays with the first view).  Fused inputs with GSPLAT_BACKWARD_SH_JACOBIAN, all six gradients given.  */
#define GSPLAT_BACKWARD_ACCUMULATE 16
int64_t gsplat_frame_bytes(int64_t n, int64_t pair_capacity, const gsplat_view* v, int32_t flags);
int gsplat_forward_deferred(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame,
                            int64_t frame_bytes, int64_t pair_capacity, void* counters, int64_t counters_bytes,
                            void* bin_scratch, int64_t bin_scratch_bytes, gsplat_counts* counts_host, void* counts_event,
                            float* image, int32_t flags, void* stream);
int gsplat_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                    int64_t pair_capacity, const float* grad_image, const gsplat_gaussian_grads* out, float* grad_logit,
                    void* det_scratch, int64_t det_scratch_bytes, int32_t flags, void* stream);

/* Data-parallel exchange helper: per view the SH-coefficient gradient is the outer product of the 3
 * colour-logit gradients with the 16 SH basis values of the view direction, so ranks exchange 12 B per Gaussian and view
 * instead of 192 B per Gaussian and rebuild the sum here:
 *   grad_f_dc[i,ch] = scale * sum_v grad_logit[v,i,ch] * Y_0,  grad_f_rest[i, ch*15 + k-1] = scale * sum_v grad_logit[v,i,ch] * Y_k(d_v(i))
 * pos[n,3]; eyes[n_views,3] = camera positions (c2w[:3,3]) on the DEVICE; grad_logit[n_views,n,3].                        */
int gsplat_sh_accumulate(int64_t n, int32_t n_views, const float* pos, const float* eyes, const float* grad_logit,
                         float scale, float* grad_f_dc, float* grad_f_rest, void* stream);
/* The same colour-logit gradients [n,3] straight from gsplat_rasterize_backward's grad2d (and the colours kept in
 * project_state), without waiting for gsplat_project_backward: the exchange of the logit gradients can overlap it.
 * (In the factored form of gsplat_project_backward out->color may then be NULL.)                                      */
int gsplat_logit_grad(int64_t n, const gsplat_view* v, const void* project_state, const float* grad2d,
                      float* grad_logit, void* stream);

/* ---- the two small exported functions as stand-alone ops --------------------------------------- */
int gsplat_build_sigma(int64_t n, const float* scale_raw, const float* q_raw, float* sigma, void* stream);
int gsplat_build_sigma_backward(int64_t n, const float* scale_raw, const float* q_raw, const float* grad_sigma,
                                float* grad_scale_raw, float* grad_q_raw, void* stream);
int gsplat_evaluate_sh(int64_t n, const float* f_dc, const float* f_rest, const float* points, const float* c2w,
                       float* color, void* stream);
int gsplat_evaluate_sh_backward(int64_t n, const float* f_dc, const float* f_rest, const float* points,
                                const float* c2w, const float* grad_color, float* grad_f_dc, float* grad_f_rest,
                                float* grad_points, void* stream);

/* ---- next row of the path: the training loss ------------------------------------
 * compute_loss() of the reference (gaussian_splatting/losses.py:158-185; l1_loss :27, ssim_loss :44,
 * 11 x 11 Gaussian window, sigma 1.5, zero padding): lambda_l1 * mean|pred - target| + lambda_ssim * (1 - SSIM).
 * pred/target/grad_pred are [batch, H, W, 3] fp32 device arrays.  values[3] (device) receives (l1, 1 - ssim, total);
 * grad_pred (nullable) receives d total / d pred.  scratch: gsplat_loss_scratch_bytes(batch, H, W, grad_pred != NULL) device
 * bytes (the partial sums, and -- with a gradient -- the three partial-derivative maps of the SSIM term, 36 B per value).   */
int64_t gsplat_loss_scratch_bytes(int64_t batch, int32_t H, int32_t W, int32_t with_grad);
int gsplat_loss(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1,
                float lambda_ssim, float* values, float* grad_pred, void* scratch, void* stream);
/* The same loss as two calls, for a caller whose graph supplies d L / d total later (an autograd node).  gsplat_loss_forward:
 * values[3] = scale * (l1, 1 - ssim, total), *total (nullable, device) = values[2]; with keep_maps the partial-derivative maps stay in
 * scratch (sized with_grad = 1) for gsplat_loss_backward, which writes grad_pred = scale * (*upstream) * d total / d pred
 * (upstream: device scalar, NULL = 1) -- the upstream factor is multiplied in by the kernel, not by a pass over the gradient.  */
int gsplat_loss_forward(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1,
                        float lambda_ssim, float scale, float* values, float* total, void* scratch, int32_t keep_maps, void* stream);
int gsplat_loss_backward(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1,
                         float lambda_ssim, float scale, const float* upstream, float* grad_pred, void* scratch, void* stream);

/* ---- next row 2: the optimiser step of scripts/train.py:394-401, 536-538 ---------------------
 * gsplat_clip_grad_norm = torch.nn.utils.clip_grad_norm_ on one tensor: coef_and_norm[2] (device) receives
 * (min(1, max_norm / (||grad|| + 1e-6)), ||grad||); nothing is scaled yet and nothing is read back to the host.
 * gsplat_adam_step = one torch.optim.Adam update (amsgrad off, no weight decay) of one flat fp32 tensor at 1-based
 * `step`; if grad_scale (device scalar, e.g. the clip coefficient) is given the gradient is multiplied by it in place
 * first, exactly like clip_grad_norm_ followed by optimizer.step().                                                  */
int64_t gsplat_clip_scratch_bytes(void);
int gsplat_clip_grad_norm(int64_t n, const float* grad, float max_norm, float* coef_and_norm, void* scratch, void* stream);
int gsplat_adam_step(int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                     float beta2, float eps, int32_t step, const float* grad_scale, void* stream);
/* The same update for up to 8 tensors in ONE launch (the six parameter groups of scripts/train.py:394-401): group k is
 * gsplat_adam_step(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale) with its own lr / step.          */
typedef struct gsplat_adam_group {
    int64_t n;
    float* param;
    float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float lr;
    int32_t step;
    const float* grad_scale;      /* nullable device scalar */
} gsplat_adam_group;
int gsplat_adam_step_multi(int32_t n_groups, const gsplat_adam_group* groups, float beta1, float beta2, float eps, void* stream);

/* gsplat_backward with the Adam step of f_rest folded into the projection backward: the 45 SH gradients of a Gaussian (192 of the 236
 * gradient bytes) are applied to f_rest as they are formed -- they are neither written nor read again by the optimiser.  For an
 * iteration of ONE view (gradients of several views must be added up before a step): f_rest->param must be the f_rest the frame
 * was rendered from, f_rest->n = 45 n, no grad_scale; out->f_rest is not written (may be NULL); every other gradient as in
 * gsplat_backward.  Needs fused inputs and GSPLAT_BACKWARD_SH_JACOBIAN (a frame queued without GSPLAT_FRAME_NO_SH_JACOBIAN); both
 * phases run.  The same arithmetic, instruction for instruction, as gsplat_adam_step on the gradient gsplat_backward would write.
 * A frame whose pairs outgrew pair_capacity (n_binned > pair_capacity: its gradients are garbage, the caller renders it again) or
 * that has nothing on screen steps NOTHING: the kernel reads the frame's device counters itself; the caller, once it has read them
 * too, knows which of the two happened (and must not count the step).                                                          */
int gsplat_backward_adam_rest(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                              int64_t pair_capacity, const float* grad_image, const gsplat_gaussian_grads* out, void* det_scratch,
                              int64_t det_scratch_bytes, int32_t flags, const gsplat_adam_group* f_rest, float beta1, float beta2,
                              float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_MI355X_H */
