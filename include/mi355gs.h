/*
 * mi355gs.h — C ABI of libmi355gs.so, the MI355X (gfx950 / CDNA4) implementation of the
 * InstantSplat train/render hot path.
 *
 * Boundary contract:
 *   - plain pointers, sizes and scalars only; no C++ / torch types cross this line;
 *   - every buffer (inputs, outputs, scratch) is allocated and owned by the CALLER (PyTorch's
 *     caching allocator in the Python binding); the library never allocates or frees device memory;
 *   - every kernel is enqueued on the `stream` argument (a hipStream_t passed as void*; the Python
 *     binding passes torch.cuda.current_stream().cuda_stream); calls return after enqueueing unless
 *     stated otherwise;
 *   - all floating point is fp32, all tensors contiguous and on the same device;
 *   - return value: 0 on success, a negative MI355GS_E* code otherwise (mi355gs_error_string());
 *   - state: the per-operator entry points keep nothing between calls.  Exceptions, all host-side and documented at their
 *     declarations: the opt-in event timing (mi355gs_profile_*), the process-wide tuning knob mi355gs_tune_min_units, and
 *     the trainer handle.  Threading: calls may be made from any host thread, one call at a time per stream; the
 *     composite entry points (mi355gs_posed_*, mi355gs_trainer_*) pass per-call context to the operator entry points
 *     they re-enter through a THREAD-LOCAL hook block (csrc/common.h, GsFusedStepHooks), set and cleared inside the call —
 *     so concurrent calls from different threads do not see each other's context, and nothing of it survives the call.
 *
 * Each entry point names the reference interface it replaces. The reference's three native
 * operators are un-vendored git submodules (reference .gitmodules:1-12), so the citations are
 * to the reference's CALL SITES, which define the signatures and semantics this ABI must serve.
 */
#ifndef MI355GS_H
#define MI355GS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355GS_ABI_VERSION 10

/* error codes */
#define MI355GS_OK 0
#define MI355GS_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported degree) */
#define MI355GS_ELAUNCH (-2)   /* a HIP launch / runtime call failed (debug mode: which kernel is logged) */
#define MI355GS_EOVERFLOW (-3) /* instance capacity too small for this frame (see mi355gs_raster_forward_render) */

int mi355gs_abi_version(void);
const char* mi355gs_error_string(int code);

/* ------------------------------------------------------------------------------------------------
 * Differentiable Gaussian rasterizer
 * replaces: diff_gaussian_rasterization._C.rasterize_gaussians / rasterize_gaussians_backward,
 *           reached through GaussianRasterizer.forward at reference gaussian_renderer/__init__.py:126-135
 *           with the settings tuple built at gaussian_renderer/__init__.py:60-76.
 *
 * Scratch layout (caller allocates `bytes` from the *_bytes() queries, 256-byte aligned):
 *   geom    : per-Gaussian records written by preprocess, read by render and backward
 *   tiles   : per-tile counters / offsets; per-pixel final transmittance and contributor counts
 *   binning : per-instance sort keys and the depth-sorted per-tile Gaussian index lists; the backward's work units
 *             (runs of 64, 128, 256 or 512 instances of a tile's list — csrc/common.h GS_SEG, gs_unit_level_for — chosen
 *             on the device from the frame's instance count; the capacity passed to the render / backward calls bounds it) and the per-pixel state the forward leaves at their boundaries
 * The forward is split in two so the caller can size `binning` exactly (one 4-byte D2H read of
 * *num_rendered between the calls, as the reference operator does internally) or skip the read
 * and pass a capacity bound (no host sync; overflow is reported through *num_rendered > capacity).
 * ---------------------------------------------------------------------------------------------- */
size_t mi355gs_raster_geom_bytes(int P);
size_t mi355gs_raster_tiles_bytes(int W, int H);
size_t mi355gs_raster_binning_bytes(int64_t num_instances, int W, int H);

/* Stage 1: per-Gaussian projection (frustum cull, EWA 2-D covariance, conic, radius, tile rect,
 * SH -> RGB), per-tile instance counts and their exclusive scan.
 *   means3D[P,3] scales[P,3] rotations[P,4] (w,x,y,z, used un-normalised) opacities[P]
 *   shs[P,M,3] (D = active degree 0..3, M = coefficients stored per Gaussian) or colors_precomp[P,3];
 *   split SH storage (the reference's own parameter layout, scene/gaussian_model.py:168-169): when shs_rest is
 *   non-null, shs is the DC coefficient [P,1,3] and shs_rest the remaining M-1 coefficients [P,M-1,3] — no
 *   cat(f_dc, f_rest) (scene/gaussian_model.py:114-117) has to be materialised
 *   cov3D_precomp[P,6] replaces scales/rotations when non-null
 *   viewmatrix[16], projmatrix[16]: row-vector convention, i.e. the transposed matrices the
 *     reference stores (scene/cameras.py:54-55) in flat memory; campos[3]
 *   radii[P] (int32, output); num_rendered: one int32 the device can write — device memory, or pinned host memory mapped into
 *   the device's address space, in which case the count reaches the host without a copy — receives the instance count R
 *   visible (ABI v7, may be null): uint8[P] (a torch.bool tensor's memory), 1 where radii > 0 — the `visibility_filter` the
 *   reference's render() computes with an elementwise kernel behind the operator (gaussian_renderer/__init__.py:142)
 *   grad_scratch (ABI v7, may be null): the mi355gs_raster_grad_scratch_bytes(P) buffer this frame's backward will be given,
 *   if the caller holds it already.  The projection kernel then clears it on its way (the per-tile counters are always
 *   cleared that way: no memset is enqueued in front of a frame), and the backward is told so: grad_scratch_is_clear = 1 */
int mi355gs_raster_forward_preprocess(
    void* stream, int P, int D, int M, int W, int H,
    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tanfovx, float tanfovy, int prefiltered,
    int32_t* radii, void* geom, void* tiles, int32_t* num_rendered, uint8_t* visible, void* grad_scratch, int debug);

/* Stage 2: scatter instances to their tiles, sort every tile's list front-to-back
 * (depth, then Gaussian index), alpha-composite.  capacity = instances `binning` was sized for.
 *   bg[3]; out_color[3,H,W] planar. */
int mi355gs_raster_forward_render(
    void* stream, int P, int W, int H, int64_t capacity, const float* bg,
    const void* geom, void* tiles, void* binning, float* out_color, int debug);

/* Stage 2 for a frame NO BACKWARD WILL FOLLOW (ABI v9) — every render under torch.no_grad(): reference render.py:87,137,177
 * (render_set, render_set_optimize's final renders, the FPS loop at :172-186), train.py:277 (training_report), evaluation.
 * Same scatter, sort and compositing, bit-identical out_color / radii / per-pixel state in `tiles`; nothing is left behind for a
 * backward: `binning` is only mi355gs_raster_binning_bytes_render_only() bytes (sort keys + the per-tile lists, 12 B per
 * instance — the training layout adds 4 KiB per 64-instance unit), and the compositing kernel's render-only instantiation stores
 * no boundary records, hit masks, unit table or quadrant maxima.  mi355gs_raster_backward must not be called on such a frame. */
size_t mi355gs_raster_binning_bytes_render_only(int64_t num_instances, int W, int H);
int mi355gs_raster_forward_render_only(
    void* stream, int P, int W, int H, int64_t capacity, const float* bg,
    const void* geom, void* tiles, void* binning, float* out_color, int debug);

/* Backward of both stages.  dL_dpix[3,H,W] in; gradients out (all written, zero where unused):
 *   dL_dmeans3D[P,3] dL_dmeans2D[P,3] (x,y in the reference's NDC-scaled screen units, z = 0)
 *   dL_dshs[P,M,3] (split storage: dL_dshs[P,1,3] + dL_dshs_rest[P,M-1,3]) or dL_dcolors[P,3], dL_dopacities[P],
 *   dL_dscales[P,3] dL_drotations[P,4] or dL_dcov3D[P,6]
 *   geom/tiles/binning/capacity/radii/out_color: exactly what the forward of this frame used and produced
 *   grad_scratch: mi355gs_raster_grad_scratch_bytes(P) bytes (per-Gaussian accumulators; the last 256 bytes, from
 *   mi355gs_raster_grad_gate_offset(P) on, are the eight gate flags mi355gs_posed_backward leaves for the optimizer)
 *   grad_scratch_is_clear (ABI v7): 1 = this frame's forward was given grad_scratch and nothing has written to it since
 *   (the backward then enqueues no memset); 0 = the backward clears it itself.  A second backward of the same frame passes 0. */
size_t mi355gs_raster_grad_scratch_bytes(int P);
size_t mi355gs_raster_grad_gate_offset(int P);
int mi355gs_raster_backward(
    void* stream, int P, int D, int M, int W, int H, const float* bg,
    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos,
    float tanfovx, float tanfovy,
    const void* geom, void* tiles, const void* binning, int64_t capacity, const int32_t* radii,
    const float* out_color, const float* dL_dpix, void* grad_scratch,
    float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dshs_rest, float* dL_dcolors, float* dL_dopacities,
    float* dL_dscales, float* dL_drotations, float* dL_dcov3D, int grad_scratch_is_clear, int debug);

/* Visibility test only — replaces diff_gaussian_rasterization._C.mark_visible
 * (GaussianRasterizer.markVisible; not called by the reference's scripts). present[P] uint8. */
int mi355gs_raster_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                                const float* projmatrix, uint8_t* present);

/* Frame statistics for roofline accounting (bench.py): stats[0] = R, the number of (tile, Gaussian)
 * instances of the last forward that used `tiles`; stats[1] = R_eff = sum over tiles of the largest
 * per-pixel contributor count, i.e. the instances the composite kernels actually had to consume
 * (SURVEY.md 8d).  stats: device int64[2]. */
int mi355gs_raster_frame_stats(void* stream, int W, int H, const void* tiles, int64_t* stats);

/* Tuning / test knob of the segmented backward (csrc/common.h, GS_MIN_UNITS): a frame's backward units are lengthened
 * (2, 4, 8 chunks of 64 instances) only while at least `min_units` of them remain.  Returns the previous value;
 * min_units <= 0 only queries.  Process-wide; the default (40960) is what every measurement uses — tests lower it to
 * run the multi-chunk path on small scenes.  It also enters the buffer-size queries: for the per-operator entry points set it
 * before sizing a frame's buffers and keep it until that frame's backward has been enqueued; a trainer handle takes a
 * snapshot of it at mi355gs_trainer_create and uses that for all its calls. */
int mi355gs_tune_min_units(int min_units);

/* Deterministic-backward mode (ABI v9; SURVEY.md 5 "determinism self-check").  The default backward adds the nine screen-space
 * moments of a Gaussian from every (tile, unit) that replays it with float atomics, in whatever order the waves arrive: two runs
 * of the same frame differ in the last bits, two runs of the same scene drift apart.  on = 1: every (Gaussian, tile) instance's
 * moments are stored to a row of their own and summed per Gaussian in the order of its tile rectangle (y outer, x inner) —
 * bit-identical gradients run to run, and with them bit-identical training.  Same arithmetic otherwise; costs five small
 * launches, a memset and 52 bytes per instance per backward (measured at C3: 0.84-0.87 x the default rate).
 * It enters the buffer-size queries (mi355gs_raster_binning_bytes: + 52 B per instance; mi355gs_raster_grad_scratch_bytes:
 * + ~4 B per Gaussian): set it before sizing a frame's buffers and keep it until that frame's backward has been enqueued; a
 * trainer handle takes a snapshot at mi355gs_trainer_create.  The work-counting instantiation (mi355gs_profile_work_counters)
 * is not available in this mode.  Returns the previous value; on < 0 only queries.  Process-wide. */
int mi355gs_tune_deterministic(int on);

/* Convention of dL_dscales under scale_modifier != 1 (ABI v9).  The covariance is built from s = scale_modifier * scale
 * (reference gaussian_renderer/__init__.py:29,66 is the only place the modifier comes from; it trains with 1.0).
 *   mode 0 (default): dL_dscales = dL/ds, the gradient with respect to the MODIFIED scale — what the published operator's
 *     backward returns (its computeCov3D backward forms s = mod * scale first and writes dL/ds into dL_dscale; recalled, the
 *     operator is an un-vendored submodule: reference .gitmodules:4-6).  "Results identical to the reference's" is the bar.
 *   mode 1: dL_dscales = scale_modifier * dL/ds, the true derivative with respect to `scales`.
 * The two are the same number at scale_modifier = 1.  Rotation gradients are not affected.  Returns the previous mode;
 * mode < 0 only queries.  Process-wide, read when a backward is enqueued. */
int mi355gs_tune_scale_grad(int mode);

/* Optional in-library kernel timing with HIP events recorded on the launch stream, so a caller that
 * cannot see the kernels (they are enqueued inside this library) can still attribute time to them.
 * kind: 0 = composite forward (training instantiation), 1 = composite backward, 2 = composite forward, render-only
 * instantiation, 3 = the fused L1 + SSIM loss pass (k_l1_ssim_fused), 4 = the per-tile sort (k_sort_tiles), 5 = the per-tile
 * instance count (k_count_tiles_lds).  profile_read synchronises the recorded events
 * and returns the summed milliseconds and launch count since profile_begin. */
int mi355gs_profile_begin(void);
/* Time only every `every`-th launch of a kind (default 1: all).  An event pair costs ~3.5 us of stream time, which matters when
 * the timed kernels are part of a measured loop.  Returns the previous period; every <= 0 only queries. */
int mi355gs_profile_set_period(int every);
/* While `counters` (device uint64[16], zeroed by the caller) is non-null, every composite launch runs its counting
 * instantiation and ADDS — backward: [0] (Gaussian, tile) steps, [1] quadrant bodies evaluated, [2] of those with at least one
 * valid pixel, [3] valid (pixel, Gaussian) pairs, [4] steps that ended in a reduction + atomics, [5] waves that did work;
 * forward: [8] staged 64-record groups (one cull each), [9] (Gaussian, quadrant) hits, [10] walk steps (two hits each),
 * [11] (pixel, Gaussian) pairs that pass the alpha tests, [12] of those blended (not the stopping one), [13] quadrant waves.
 * null switches back to the shipped kernels (which have no counters).  Process-wide, measurement only. */
int mi355gs_profile_work_counters(void* counters);
int mi355gs_profile_read(int kind, double* total_ms, int* launches);
int mi355gs_profile_end(void);
/* roctx ranges (ABI v10; SURVEY.md section 5, row 1: the reference has no ranges at all, its profile is TensorBoard's iter_time,
 * train.py:114-115,140,178).  While on, every entry point that enqueues kernels brackets itself with
 * roctxRangePushA(<its name>) / roctxRangePop() — the stages of a train step appear as nested ranges under
 * "mi355gs_trainer_step" in `rocprofv3 --marker-trace --kernel-trace` (tools/prof.sh).  The roctx library
 * (librocprofiler-sdk-roctx.so, else libroctx64.so) is opened with dlopen on first use: the library has no link-time dependency
 * on it, and without it this stays off.  on = 1 / 0: switch (also on at load with MI355GS_ROCTX=1); on = -1: query.
 * Returns the number of ranges pushed since the library was loaded (>= 0), or MI355GS_EINVAL if roctx could not be opened when
 * asked to switch on. */
int mi355gs_profile_ranges(int on);

/* ------------------------------------------------------------------------------------------------
 * fused SSIM (+ optional L1) loss
 * replaces: fused_ssim.fused_ssim(img1, img2) at reference train.py:173 (same value as the
 *           reference's own fallback utils/loss_utils.py:55-85: 11x11 Gaussian window, sigma 1.5,
 *           zero "same" padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements).
 *   img1, img2 [B,C,H,W]; scratch: mi355gs_ssim_scratch_bytes() bytes (per-workgroup partial sums,
 *   reduced in a fixed order so the result is run-to-run deterministic)
 *   ssim_mean / l1_mean: device float[1] each (either may be null): mean SSIM map, mean |img1-img2|
 *   dm_dmu1, dm_dsigma1_sq, dm_dsigma12 [B,C,H,W]: saved partials for backward (all null = inference)
 *   padding_valid: 0 = fused_ssim(padding="same") (the reference's use); 1 = padding="valid": the map is computed the same
 *   way but only the region where the window lies inside the image (5 px in from every edge) is averaged and passes
 *   gradient (l1_mean / l1_grad_scale must then be null; H, W > 10)
 * ---------------------------------------------------------------------------------------------- */
size_t mi355gs_ssim_scratch_bytes(int B, int C, int H, int W);
int mi355gs_ssim_forward(void* stream, int B, int C, int H, int W, const float* img1, const float* img2,
                         float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
                         void* scratch, float* ssim_mean, float* l1_mean, int padding_valid);
/* dL_dimg1 = ssim_grad_scale * d(ssim_mean)/dimg1 + l1_grad_scale * d(l1_mean)/dimg1
 * (scales are read from device scalars so no host sync is needed; null = 0). */
int mi355gs_ssim_backward(void* stream, int B, int C, int H, int W, const float* img1, const float* img2,
                          const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                          const float* ssim_grad_scale, const float* l1_grad_scale, float* dL_dimg1, int padding_valid);
/* The reference's training loss (train.py:171-176), loss = (1 - lambda) * L1 + lambda * (1 - SSIM), and dloss_dimg1 [B,C,H,W]
 * (the gradient for dL/dloss = 1) from ONE pass over the two images: the SSIM partial-derivative maps never leave LDS, about a
 * third of the traffic of mi355gs_ssim_forward + _backward.  ssim_mean / l1_mean (optional) and loss: device float[1] each.
 * The binding's forward calls this and keeps the gradient; its backward multiplies it by the incoming dL/dloss. */
int mi355gs_l1_ssim_loss_fused(void* stream, int B, int C, int H, int W, const float* img1, const float* img2, void* scratch,
                               float lambda_dssim, float* ssim_mean, float* l1_mean, float* loss, float* dloss_dimg1);

/* ------------------------------------------------------------------------------------------------
 * l1_loss(network_output, gt) = mean |a - b| over n contiguous floats (reference utils/loss_utils.py:39-40, called at
 * train.py:171) and its gradient w.r.t. a: sgn(a - b) * (*grad_scale) / n, as autograd computes it for abs(a - b).mean()
 * (ABI v8).  scratch: mi355gs_l1_scratch_bytes(n) bytes of device memory; mean_out, grad_scale: device float[1].
 * Two launches forward (per-workgroup partial sums in float, finished in double in a fixed order), one backward.
 * ---------------------------------------------------------------------------------------------- */
size_t mi355gs_l1_scratch_bytes(int64_t n);
int mi355gs_l1_loss_forward(void* stream, int64_t n, const float* a, const float* b, void* scratch, float* mean_out);
int mi355gs_l1_loss_backward(void* stream, int64_t n, const float* a, const float* b, const float* grad_scale, float* d_a);
/* ------------------------------------------------------------------------------------------------
 * The reference's loss EXPRESSION as written (ABI v9) — reference train.py:171-176:
 *     Ll1 = l1_loss(image, gt_image);  ssim_value = fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
 *     loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim_value);  loss.backward()
 * two operator calls on the same pair of images and four scalar operations on 0-dim tensors: sixteen eager launches
 * forward + backward.  These entry points let a binding serve that source text unmodified in three launches — TWO when the
 * backward follows the materialisation directly (program_eval_grad) —
 * (instantsplat_amd/loss_utils.py, lazy_loss.py: l1_loss runs the pair forward, fused_ssim on the same tensors takes its other half from
 * it, the scalar arithmetic is recorded on the host and evaluated by one launch, the backward of the whole expression is one
 * launch over the image).
 *   pair_forward: ONE pass over the images, one launch (the kernel of mi355gs_l1_ssim_loss_fused) -> dssim_dimg1[B,C,H,W] =
 *     d(ssim_mean)/d(img1), and both means as per-workgroup partial sums in `scratch` (mi355gs_ssim_scratch_bytes(); it must
 *     stay untouched until the last program_eval of the pair).
 *   program_eval: one launch that FINISHES the two means from `scratch` (fixed order, double accumulation: the bits
 *     mi355gs_ssim_forward / mi355gs_l1_ssim_loss_fused report) into *ssim_mean / *l1_mean (device float[1] each) and evaluates a
 *     postfix program of n_ops <= MI355GS_LOSS_PROGRAM_MAX operations over them, ops[i] / consts[i] HOST arrays, in float32 with
 *     one rounding per operation (what eager PyTorch computes for the same expression):
 *     L1 / SSIM push a mean; MULK x*k, ADDK x+k, RSUBK k-x, DIVK x*(1/k), NEG -x act on the top of the stack; ADD / SUB pop two
 *     (a b -> a+b, a-b).  *out = the one value left.  A program that underflows the stack or leaves more than one value is refused.
 *     host_out (may be null; 8-byte aligned float[2]): a second destination the DEVICE can write and the HOST can read without a
 *     copy — pinned host memory mapped into the device's address space: the kernel stores (value, ticket) there as one 8-byte
 *     store.  A caller that presets the slot and polls for its ticket has the value as soon as this launch has run — reference
 *     train.py:188 `loss.item()` then neither enqueues a copy nor waits for the backward kernels queued behind the loss.
 *   pair_backward: d_img1[n] = ((g_l1 ? *g_l1 : 1) * c_l1 / n) * sgn(img1 - img2) + ((g_ssim ? *g_ssim : 1) * c_ssim) * dssim_dimg1
 *     — the gradient of  c_l1 * l1_mean + c_ssim * ssim_mean  scaled by incoming gradients read from device scalars (autograd's own
 *     operation order for abs(a - b).mean()); dssim_dimg1 may be null when c_ssim == 0.
 * ---------------------------------------------------------------------------------------------- */
#define MI355GS_LOSS_PROGRAM_MAX 16
#define MI355GS_LOSS_OP_L1 0
#define MI355GS_LOSS_OP_SSIM 1
#define MI355GS_LOSS_OP_MULK 2
#define MI355GS_LOSS_OP_ADDK 3
#define MI355GS_LOSS_OP_RSUBK 4
#define MI355GS_LOSS_OP_DIVK 5
#define MI355GS_LOSS_OP_NEG 6
#define MI355GS_LOSS_OP_ADD 7
#define MI355GS_LOSS_OP_SUB 8
int mi355gs_l1_ssim_pair_forward(void* stream, int B, int C, int H, int W, const float* img1, const float* img2, void* scratch,
                                 float* dssim_dimg1);
int mi355gs_l1_ssim_pair_backward(void* stream, int64_t n, const float* img1, const float* img2, const float* dssim_dimg1,
                                  const float* g_l1, float c_l1, const float* g_ssim, float c_ssim, float* d_img1);
int mi355gs_loss_program_eval(void* stream, int n_ops, const int32_t* ops, const float* consts, int B, int C, int H, int W,
                              const void* scratch, float* ssim_mean, float* l1_mean, float* out, float* host_out, float ticket);
/* program_eval AND pair_backward for dL/d(value) = 1 in ONE launch (g_l1 = g_ssim = 1; c_l1 / c_ssim: the partial derivatives of
 * the program with respect to the two means): what `loss.backward()` right behind the materialisation needs — the gradient over
 * the image does not depend on the means, so one workgroup finishes the value while the others write d_img1[B*C*H*W]. */
int mi355gs_loss_program_eval_grad(void* stream, int n_ops, const int32_t* ops, const float* consts, int B, int C, int H, int W,
                                   const void* scratch, float* ssim_mean, float* l1_mean, float* out, float* host_out, float ticket,
                                   const float* img1, const float* img2, const float* dssim_dimg1, float c_l1, float c_ssim, float* d_img1);

/* ------------------------------------------------------------------------------------------------
 * simple-knn
 * replaces: simple_knn._C.distCUDA2(points) at reference scene/gaussian_model.py:156 —
 *           mean of the squared distances to the 3 nearest other points (exact).
 * ---------------------------------------------------------------------------------------------- */
size_t mi355gs_knn_scratch_bytes(int N);
int mi355gs_knn_dist2(void* stream, int N, const float* points, float* mean_dist2, void* scratch);

/* ------------------------------------------------------------------------------------------------
 * per-point Adam
 * replaces: PerPointAdam.step at reference scene/per_point_adam.py:34-100 for one parameter tensor
 *   n = elements, row = elements per point (per_point_lr has n/row entries, or null)
 *   grad_sumsq: device float[1] holding sum(grad^2) — the reference gates the moment update on the
 *   whole-tensor norm being > 0 (per_point_adam.py:62-69); pass null to always update.
 * ---------------------------------------------------------------------------------------------- */
int mi355gs_adam_step(void* stream, int64_t n, int row, float* param, const float* grad, float* exp_avg,
                      float* exp_avg_sq, const float* per_point_lr, const float* grad_sumsq,
                      float lr, float beta1, float beta2, float eps, int step);

/* All parameter tensors of one optimizer step in two launches (<= 8 tensors per call).  The arrays are
 * HOST arrays of length ntensors holding device pointers / per-tensor scalars; per_point_lr[t] may be null;
 * step[t] is the 1-based step count of tensor t; scratch: device float[8].  The whole-tensor gradient gate
 * of the reference (per_point_adam.py:62-69) is evaluated on the device from the summed squares — or, when `gate` is
 * non-null, taken from flags the gradients' producer has already left on the device: tensor t updates its moments iff
 * gate[gate_index[t]] > 0 (device float[8] / host int32[ntensors], entries 0..7; `scratch` may then be null) and the
 * pass over all gradients is not launched.  mi355gs_posed_backward leaves such flags behind its gradient records
 * (group order xyz, f_dc, f_rest, opacity, scaling, rotation, pose); the caller vouches that grads[t] is exactly the
 * tensor that call wrote.
 * live / seq (ABI v7, live may be null): a memory of gated-off tensors ACROSS calls, owned by the caller — device uint32[16],
 * zeroed once, handed to every step of the same tensor list together with a sequence number that differs from call to call
 * and is never 0.  A tensor whose gradient is all zero keeps its moments and takes the parameter step p - s * (m / denom),
 * which is the identity wherever the first moment is zero; a call that scanned such a tensor completely without meeting a
 * non-zero first moment says so in `live`, and later calls skip the tensor without reading anything (f_rest before the SH
 * degree is raised: 4 B per element per step otherwise).  Any update of the tensor resets its entry.  The caller must zero
 * `live` again if anything but these calls writes the moments. */
int mi355gs_adam_multi_step(void* stream, int ntensors, const int64_t* numel, const int32_t* row, float* const* params,
                            const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            const float* const* per_point_lr, const float* lr, float beta1, float beta2, float eps,
                            const int32_t* step, float* scratch, const float* gate, const int32_t* gate_index,
                            uint32_t* live, uint32_t seq);

/* ------------------------------------------------------------------------------------------------
 * InstantSplat camera-frame transform fused with the Gaussian activations
 * replaces: the PyTorch ops of reference gaussian_renderer/__init__.py:81-103 —
 *   rel_w2c = get_camera_from_tensor(pose) (utils/pose_utils.py:57-84, quaternion normalised),
 *   means_cam = rel_w2c @ [xyz,1], rot_cam = quadmultiply(pose[:4], rotation) (raw quaternions),
 *   opacity = sigmoid(_opacity), scales = exp(_scaling) (scene/gaussian_model.py:101-124) — and their
 *   autograd backward, including the reduction over all Gaussians to dL/dpose[7].
 *   pose[7] = (qw,qx,qy,qz,tx,ty,tz) on the device; scratch: device float[32] (16 sums, a ticket counter, padding).
 * ---------------------------------------------------------------------------------------------- */
int mi355gs_pose_forward(void* stream, int P, const float* xyz, const float* rot, const float* scaling,
                         const float* opacity_logit, const float* pose, float* means_cam, float* rot_cam, float* scales,
                         float* opac);
int mi355gs_pose_backward(void* stream, int P, const float* xyz, const float* rot, const float* scales, const float* opac,
                          const float* pose, const float* g_means, const float* g_rot, const float* g_scales,
                          const float* g_opac, float* d_xyz, float* d_rot, float* d_scaling, float* d_opacity_logit,
                          float* d_pose, float* scratch);

/* ------------------------------------------------------------------------------------------------
 * InstantSplat's render() as the operator sees it: Gaussians in the WORLD frame with RAW parameters plus the learnable
 * camera pose — the projection kernels apply the camera-frame transform and the activations themselves, and their backward
 * goes all the way to the raw-parameter gradients and dL/dpose[7]: no camera-frame intermediates in HBM, three launches fewer
 * per iteration than mi355gs_pose_forward + mi355gs_raster_* + mi355gs_pose_backward, ONE autograd node in the binding.
 * replaces: everything reference gaussian_renderer/__init__.py:81-135 does between the GaussianModel and the image for the
 *   default pipeline (SH colours, scale/rotation covariance): get_camera_from_tensor, the [4,4]@[4,P] transform,
 *   quadmultiply, sigmoid / exp, cat(f_dc, f_rest), GaussianRasterizer.forward — and their autograd.
 *   xyz[P,3], f_dc[P,1,3], f_rest[P,15,3] (may be null while D == 0: only the DC coefficient is read), opacity_logit[P,1],
 *   log_scales[P,3], rotation[P,4] raw (w,x,y,z), pose[7] = (qw,qx,qy,qz,tx,ty,tz), all on the device.
 *   view_identity[16] / origin[3]: device constants (identity matrix, zeros) — InstantSplat hands the operator an
 *   identity view and a camera at the origin (reference :55-59); kept as arguments because the library owns no memory.
 *   Stage 2 of the forward is mi355gs_raster_forward_render, unchanged.  visible / grad_scratch / grad_scratch_is_clear: as for
 *   mi355gs_raster_forward_preprocess / _backward.
 *   backward: pose_scratch = device float[16 * ((P + 255) / 256) + 32]; d_f_rest may be null while D == 0 (the reference's
 *   gradient for it is all zero then); d_* receive dL/d(raw parameter), d_pose[7] dL/dpose.
 *   pose_rows / pose_row (ABI v8): with pose_rows > 0, d_pose is the gradient of the WHOLE pose table the 7-vector was taken
 *   from — float[pose_rows, 7] (reference scene/gaussian_model.py:134-136, `P[idx]`): row pose_row receives dL/dpose and every
 *   other row is written 0, which is what autograd's backward of the row selection builds with a fill and a copy of its own.
 *   pose_rows == 0: d_pose[7] as before.
 *   Also written: PerPointAdam's whole-tensor gate flags for these gradients, float[8] at grad_scratch +
 *   mi355gs_raster_grad_gate_offset(P): flag k > 0 <=> the gradient of group k (xyz, f_dc, f_rest, opacity, scaling,
 *   rotation, pose) has a non-zero element — what mi355gs_adam_multi_step(gate=...) consumes instead of re-reading them.
 * ---------------------------------------------------------------------------------------------- */
int mi355gs_posed_forward_preprocess(void* stream, int P, int D, int W, int H, const float* xyz, const float* f_dc,
                                     const float* f_rest, const float* opacity_logit, const float* log_scales, float scale_modifier,
                                     const float* rotation, const float* pose, const float* view_identity, const float* projmatrix,
                                     const float* origin, float tanfovx, float tanfovy, int32_t* radii, void* geom, void* tiles,
                                     int32_t* num_rendered, uint8_t* visible, void* grad_scratch, int debug);
int mi355gs_posed_backward(void* stream, int P, int D, int W, int H, const float* bg, const float* xyz, const float* f_dc,
                           const float* f_rest, const float* opacity_logit, const float* log_scales, float scale_modifier,
                           const float* rotation, const float* pose, const float* view_identity, const float* projmatrix,
                           const float* origin, float tanfovx, float tanfovy, const void* geom, void* tiles, const void* binning,
                           int64_t capacity, const int32_t* radii, const float* out_color, const float* dL_dpix,
                           void* grad_scratch, float* pose_scratch, float* d_xyz, float* d_means2D, float* d_f_dc, float* d_f_rest,
                           float* d_opacity_logit, float* d_log_scales, float* d_rotation, float* d_pose, int pose_rows, int pose_row,
                           int grad_scratch_is_clear, int debug);

/* ------------------------------------------------------------------------------------------------
 * Whole train iteration in one call
 * replaces: the body of the loop at reference train.py:140-211 — render(camera_pose=P[view]) -> (1-l)*L1 +
 *   l*(1-SSIM) -> backward -> PerPointAdam.step() — for the configuration the reference's scripts run: SH colours
 *   (active degree sh_degree 0..3 over 16 stored coefficients), scale/rotation covariance, --pp_optimizer
 *   --optim_pose.  11 launches, no host sync.
 *   Parameter tensors use the reference's GaussianModel layouts (scene/gaussian_model.py:166-171): xyz[P,3],
 *   f_dc[P,1,3], f_rest[P,15,3], opacity[P,1], scaling[P,3], rotation[P,4], poses[V,7]; exp_avg/exp_avg_sq: host
 *   arrays of 7 device pointers in the optimizer's group order (xyz, f_dc, f_rest, opacity, scaling, rotation, pose).
 *   workspace: mi355gs_trainer_workspace_bytes() bytes, owned by the caller, must outlive the handle.
 *   The handle is the only writer of the parameter and moment tensors while it is alive (it remembers across steps that
 *   a tensor without gradients has an all-zero first moment and then skips it; create a new handle after changing them).
 *   capacity: instance capacity of the binning buffers; *num_rendered (device-writable, see above) receives the true count of every
 *   step — a step with num_rendered > capacity dropped instances and must be discarded by the caller.  With
 *   do_optimizer_step = 1 the optimizer launch of such a step discards ITSELF: it reads the count and the capacity on the
 *   device and writes no parameter and no moment when the count is larger (commit gate), so the whole iteration can be enqueued
 *   before the host has seen the count and an overflowed one is simply repeated with larger buffers; its loss_out / gradients are
 *   those of the truncated frame.  The gate is STICKY (ABI v8): once a step of a handle has discarded itself, every later
 *   do_optimizer_step = 1 step of that handle discards itself too, whatever its own count — so a caller that enqueues many steps
 *   ahead of its reads of the counts finds parameters and moments exactly as the last step BEFORE the first overflow left them,
 *   needs no snapshot to return to, and continues from there with larger buffers, i.e. a new handle (ABI v9 dropped
 *   mi355gs_trainer_rearm: no caller keeps a handle whose buffers a frame has outgrown).  (mi355gs_trainer_optimizer_step, below,
 *   is gated only if asked to be.)
 *   lr[7], step[7] (1-based Adam step of each group): host arrays.  loss_out: device float[1].
 * ---------------------------------------------------------------------------------------------- */
size_t mi355gs_trainer_workspace_bytes(int P, int W, int H, int V, int64_t capacity);
void* mi355gs_trainer_create(int P, int W, int H, int V, int64_t capacity, float* xyz, float* f_dc, float* f_rest, float* opacity,
                             float* scaling, float* rotation, float* poses, float* const* exp_avg, float* const* exp_avg_sq,
                             const float* per_point_lr, void* workspace);
int mi355gs_trainer_step(void* trainer, void* stream, int view, int sh_degree, const float* gt_image, const float* projmatrix,
                         float tanfovx, float tanfovy, const float* bg, const float* lr, const int32_t* step, float beta1, float beta2, float eps,
                         float lambda_dssim, int do_optimizer_step, float* loss_out, int32_t* num_rendered_out);
/* PerPointAdam over all 7 groups with the gradients left by the last mi355gs_trainer_step(..., do_optimizer_step = 0):
 * lets a caller inspect the loss / instance count of an iteration before committing its update (commit_gate = 0: the caller's
 * own decision, not gated) — or, commit_gate = 1 (ABI v8), split an iteration in two enqueues without having seen the count:
 * the launch then carries the same sticky device-side gate as a do_optimizer_step = 1 step, on the count of the frame whose
 * gradients it applies.  (What that buys: forward + backward of iteration t + 1 can be enqueued BEFORE the host reads the loss
 * of iteration t, its update after — the device never waits for the host and the host still sees every loss.) */
int mi355gs_trainer_optimizer_step(void* trainer, void* stream, const float* lr, const int32_t* step, float beta1, float beta2,
                                   float eps, int commit_gate);
void mi355gs_trainer_destroy(void* trainer);
/* Device pointer of the gradient buffer of parameter group k (0..6: xyz, f_dc, f_rest, opacity, scaling, rotation, poses)
 * as left by the last mi355gs_trainer_step — for tests and diagnostics (gradient parity of the fused step). */
const float* mi355gs_trainer_grad(void* trainer, int k);

#ifdef __cplusplus
}
#endif
#endif /* MI355GS_H */
