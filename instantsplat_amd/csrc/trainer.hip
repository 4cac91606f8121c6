// One full InstantSplat train iteration behind ONE library call (whole-iteration enqueue once
// the host read-backs are gone).  Body of reference train.py:140-211 for the configuration the reference's
// scripts run (SH degree 0 during the first 1000 iterations, --pp_optimizer --optim_pose, scale/rotation
// covariance, SH colours):
//
//   pose transform + activations -> projection -> tile binning -> composite
//   -> (1-l)*L1 + l*(1-SSIM) -> SSIM/L1 backward -> composite backward -> projection backward
//   -> pose/activation backward (incl. the 7 pose gradients) -> PerPointAdam over all 7 parameter groups
//
// 11 kernel launches (projection, tile count, tile scan, scatter, tile sort, composite, fused loss, composite backward,
// projection backward, pose finish + loss value, PerPointAdam), no host synchronisation, no temporary allocation: every buffer lives in one caller-provided
// workspace that the trainer carves up once.  The same kernels (and launch helpers) as the op-by-op path are used,
// so results are identical up to the order of float atomics.  The instance buffers have a fixed capacity; the
// true count is written to *num_rendered every step so the caller can verify it asynchronously.
#include <stdlib.h>
#include <string.h>
#include "common.h"

int gs_loss_fused(hipStream_t, int, int, int, const float*, const float*, float, float*, void*);
int gs_loss_fused_nblocks(int, int, int);
int gs_launch_pose_finish_partials(hipStream_t, const float*, const float*, int, float*, float*, const float*, int, double, float, float*, int,
                                   int);

namespace {

struct Trainer {
  int P, W, H, V;
  int64_t capacity;
  // parameters, optimizer state (caller-owned)
  float *xyz, *f_dc, *f_rest, *opacity, *scaling, *rotation, *poses;
  float* m[7];
  float* v[7];
  const float* pplr;
  // workspace slices
  char *geom, *tiles, *binning, *grad_scratch, *ssim_scratch;
  float *image, *dL_dimg;
  int32_t* radii;
  float *g_means2D, *g_colors;
  float *g_xyz, *g_rot, *g_scaling, *g_opacity, *g_fdc, *g_frest, *g_poses;
  float *pose_scratch, *pose_partial, *adam_scratch, *consts;  // consts: identity view [16], campos [3]
  uint32_t* adam_live;  // see MultiAdamArgs::live
  uint32_t adam_seq;
  bool consts_ready;
  int det;         // the deterministic-backward knob at the same moment (it enters the layouts too)
  int min_units;   // the unit-length knob as it stood when the workspace was carved: every later call of the handle sizes and
                   // launches with THIS value, whatever mi355gs_tune_min_units has been set to since (the buffers were laid
                   // out for it)
};

struct Carver {
  char* base;
  size_t off = 0;
  template <class T> T* take(size_t n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += gs_align(n * sizeof(T));
    return p;
  }
};

size_t carve(Trainer& t, void* workspace) {
  Carver c{(char*)workspace};
  const size_t P = (size_t)(t.P > 0 ? t.P : 1), npix = (size_t)t.W * t.H;
  t.geom = c.take<char>(mi355gs_raster_geom_bytes(t.P));
  t.tiles = c.take<char>(mi355gs_raster_tiles_bytes(t.W, t.H));
  t.binning = c.take<char>(mi355gs_raster_binning_bytes(t.capacity, t.W, t.H));
  t.grad_scratch = c.take<char>(mi355gs_raster_grad_scratch_bytes(t.P));
  t.ssim_scratch = c.take<char>(mi355gs_ssim_scratch_bytes(1, 3, t.H, t.W));
  t.image = c.take<float>(3 * npix); t.dL_dimg = c.take<float>(3 * npix);
  t.radii = c.take<int32_t>(P);
  t.g_means2D = c.take<float>(3 * P); t.g_colors = c.take<float>(3 * P);
  t.g_xyz = c.take<float>(3 * P); t.g_rot = c.take<float>(4 * P); t.g_scaling = c.take<float>(3 * P); t.g_opacity = c.take<float>(P);
  t.g_fdc = c.take<float>(3 * P); t.g_frest = c.take<float>(45 * P); t.g_poses = c.take<float>(7 * (size_t)t.V);
  t.pose_scratch = c.take<float>(32); t.pose_partial = c.take<float>(16 * ((P + 255) / 256)); t.adam_scratch = c.take<float>(8); t.consts = c.take<float>(32);
  t.adam_live = c.take<uint32_t>(16);
  return c.off;
}

// commit_gate: the launch is part of the step that produced the gradients and has not been cleared by the host — it must
// turn itself into a no-op when that frame's instance count exceeded the buffers (see GsFusedStepHooks::commit_count)
int trainer_adam(Trainer* t, hipStream_t stream, const float* lr, const int32_t* step, float beta1, float beta2, float eps,
                 bool commit_gate) {
  const int P = t->P;
  const int64_t numel[7] = {3LL * P, 3LL * P, 45LL * P, (int64_t)P, 3LL * P, 4LL * P, 7LL * t->V};
  const int32_t row[7] = {3, 1, 1, 1, 1, 1, 1};
  float* params[7] = {t->xyz, t->f_dc, t->f_rest, t->opacity, t->scaling, t->rotation, t->poses};
  const float* grads[7] = {t->g_xyz, t->g_fdc, t->g_frest, t->g_opacity, t->g_scaling, t->g_rot, t->g_poses};
  const float* pplr[7] = {t->pplr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (++t->adam_seq == 0u) t->adam_seq = 1u;
  g_fused.adam_live = t->adam_live; g_fused.adam_seq = t->adam_seq;
  if (commit_gate) {
    const TilesLayout tl(t->W, t->H);
    g_fused.commit_count = (const uint32_t*)(t->tiles + tl.start) + tl.T;   // tile_start[T]: the frame's instance count
    g_fused.commit_capacity = (unsigned long long)t->capacity;
    g_fused.commit_poison = t->adam_live + 15;   // (words 0..13 are MultiAdamArgs::live's; cleared with them before the first step)
  }
  return mi355gs_adam_multi_step(stream, 7, numel, row, params, grads, t->m, t->v, pplr, lr, beta1, beta2, eps, step, t->adam_scratch, nullptr, nullptr, nullptr, 0u);
}

__global__ void k_trainer_consts(float* consts) {
  const int i = threadIdx.x;
  if (i < 16) consts[i] = (i % 5 == 0) ? 1.f : 0.f;  // identity view matrix
  else if (i < 19) consts[i] = 0.f;                   // camera position
}

}  // namespace

extern "C" {

// ---- render() with the pose inside the operator (include/mi355gs.h): the same posed projection kernels the one-call step
// uses, reached through the same thread-local hook, as two stateless entry points for the autograd binding.
int mi355gs_posed_forward_preprocess(void* stream, int P, int D, int W, int H, const float* xyz, const float* f_dc,
                                     const float* f_rest, const float* opacity_logit, const float* log_scales, float scale_modifier,
                                     const float* rotation, const float* pose, const float* view_identity, const float* projmatrix,
                                     const float* origin, float tanfovx, float tanfovy, int32_t* radii, void* geom, void* tiles,
                                     int32_t* num_rendered, uint8_t* visible, void* grad_scratch, int debug) {
  GS_RANGE();
  if (!pose || D < 0 || D > 3 || (D > 0 && !f_rest)) return MI355GS_EINVAL;
  struct Scope { ~Scope() { g_fused = GsFusedStepHooks(); } } scope;
  g_fused.posed.pose = pose;
  return mi355gs_raster_forward_preprocess(stream, P, D, D == 0 ? 1 : 16, W, H, xyz, f_dc, D == 0 ? nullptr : f_rest, nullptr,
                                           opacity_logit, log_scales, scale_modifier, rotation, nullptr, view_identity, projmatrix,
                                           origin, tanfovx, tanfovy, 0, radii, geom, tiles, num_rendered, visible, grad_scratch, debug);
}

int mi355gs_posed_backward(void* stream_, int P, int D, int W, int H, const float* bg, const float* xyz, const float* f_dc,
                           const float* f_rest, const float* opacity_logit, const float* log_scales, float scale_modifier,
                           const float* rotation, const float* pose, const float* view_identity, const float* projmatrix,
                           const float* origin, float tanfovx, float tanfovy, const void* geom, void* tiles, const void* binning,
                           int64_t capacity, const int32_t* radii, const float* out_color, const float* dL_dpix,
                           void* grad_scratch, float* pose_scratch, float* d_xyz, float* d_means2D, float* d_f_dc, float* d_f_rest,
                           float* d_opacity_logit, float* d_log_scales, float* d_rotation, float* d_pose, int pose_rows, int pose_row,
                           int grad_scratch_is_clear, int debug) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  if (!pose || !pose_scratch || !d_pose || D < 0 || D > 3 || (D > 0 && (!f_rest || !d_f_rest))) return MI355GS_EINVAL;
  if (pose_rows < 0 || (pose_rows > 0 && (pose_row < 0 || pose_row >= pose_rows))) return MI355GS_EINVAL;
  if (P <= 0)
    return hipMemsetAsync(d_pose, 0, (size_t)(pose_rows > 0 ? pose_rows : 1) * 7 * sizeof(float), stream) == hipSuccess ? MI355GS_OK : MI355GS_ELAUNCH;
  const int rows = (P + 255) / 256;
  float* gate = (float*)((char*)grad_scratch + mi355gs_raster_grad_gate_offset(P));
  int rc;
  {
    struct Scope { ~Scope() { g_fused = GsFusedStepHooks(); } } scope;
    g_fused.posed.pose = pose;
    g_fused.posed.acc = pose_scratch + (size_t)16 * rows;   // unused with `partial` set; kept valid
    g_fused.posed.partial = pose_scratch;                   // one row of 16 pose sums per projection workgroup
    // PerPointAdam's whole-tensor gate flags, in the optimizer's group order, from the kernels that write the gradients
    g_fused.gate = gate; g_fused.gate_tail = true;
    g_fused.gate_xyz = 0; g_fused.gate_sh = 1; g_fused.gate_sh_rest = 2; g_fused.gate_opacity = 3; g_fused.gate_scaling = 4; g_fused.gate_rot = 5;
    rc = mi355gs_raster_backward(stream, P, D, D == 0 ? 1 : 16, W, H, bg, xyz, f_dc, D == 0 ? nullptr : f_rest, nullptr, opacity_logit,
                                 log_scales, scale_modifier, rotation, nullptr, view_identity, projmatrix, origin, tanfovx, tanfovy,
                                 geom, tiles, binning, capacity, radii, out_color, dL_dpix, grad_scratch, d_xyz, d_means2D, d_f_dc,
                                 D == 0 ? nullptr : d_f_rest, nullptr, d_opacity_logit, d_log_scales, d_rotation, nullptr,
                                 grad_scratch_is_clear, debug);
  }
  if (rc) return rc;
  GS_KRANGE("pose_finish");
  gs_launch_pose_finish_partials(stream, pose, pose_scratch, rows, d_pose, gate + 6, nullptr, 0, 0.0, 0.f, nullptr, pose_rows, pose_row);
  GS_CHECK_LAUNCH("pose_finish");
  return MI355GS_OK;
}

size_t mi355gs_trainer_workspace_bytes(int P, int W, int H, int V, int64_t capacity) {
  if (P < 0 || W <= 0 || H <= 0 || V <= 0 || capacity < 0) return 0;
  Trainer t;
  memset(&t, 0, sizeof(t));
  t.P = P; t.W = W; t.H = H; t.V = V; t.capacity = capacity;
  return carve(t, nullptr);
}

void* mi355gs_trainer_create(int P, int W, int H, int V, int64_t capacity, float* xyz, float* f_dc, float* f_rest, float* opacity,
                             float* scaling, float* rotation, float* poses, float* const* exp_avg, float* const* exp_avg_sq,
                             const float* per_point_lr, void* workspace) {
  if (P <= 0 || W <= 0 || H <= 0 || V <= 0 || capacity <= 0 || !workspace || !exp_avg || !exp_avg_sq) return nullptr;
  if (!xyz || !f_dc || !f_rest || !opacity || !scaling || !rotation || !poses) return nullptr;
  Trainer* t = (Trainer*)calloc(1, sizeof(Trainer));
  if (!t) return nullptr;
  t->P = P; t->W = W; t->H = H; t->V = V; t->capacity = capacity;
  t->xyz = xyz; t->f_dc = f_dc; t->f_rest = f_rest; t->opacity = opacity; t->scaling = scaling; t->rotation = rotation; t->poses = poses;
  for (int k = 0; k < 7; ++k) {
    if (!exp_avg[k] || !exp_avg_sq[k]) { free(t); return nullptr; }
    t->m[k] = exp_avg[k]; t->v[k] = exp_avg_sq[k];
  }
  t->pplr = per_point_lr;
  t->min_units = gs_min_units();
  t->det = gs_deterministic();
  carve(*t, workspace);
  t->consts_ready = false;
  return t;
}

void mi355gs_trainer_destroy(void* handle) { free(handle); }

const float* mi355gs_trainer_grad(void* handle, int k) {
  Trainer* t = (Trainer*)handle;
  if (!t || k < 0 || k > 6) return nullptr;
  const float* g[7] = {t->g_xyz, t->g_fdc, t->g_frest, t->g_opacity, t->g_scaling, t->g_rot, t->g_poses};
  return g[k];
}

int mi355gs_trainer_step(void* handle, void* stream_, int view, int sh_degree, const float* gt_image, const float* projmatrix,
                         float tanfovx, float tanfovy, const float* bg, const float* lr, const int32_t* step, float beta1, float beta2, float eps,
                         float lambda_dssim, int do_optimizer_step, float* loss_out, int32_t* num_rendered_out) {
  GS_RANGE();
  Trainer* t = (Trainer*)handle;
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (!t || view < 0 || view >= t->V || sh_degree < 0 || sh_degree > 3 || !gt_image || !projmatrix || !bg || !lr || !step || !loss_out || !num_rendered_out)
    return MI355GS_EINVAL;
  const int P = t->P, W = t->W, H = t->H;
  if (!t->consts_ready) {
    GS_KRANGE("trainer_consts");
    hipLaunchKernelGGL(k_trainer_consts, dim3(1), dim3(64), 0, stream, t->consts);
    GS_CHECK_LAUNCH("trainer_consts");
    // at SH degree 0 f_rest receives no gradient: its (all-zero) gradient buffer is written once here and first
    // touched again when the degree is raised (the backward then rewrites every element each step)
    if (hipMemsetAsync(t->g_frest, 0, (size_t)P * 45 * sizeof(float), stream) != hipSuccess) return MI355GS_ELAUNCH;
    if (hipMemsetAsync(t->adam_live, 0, 16 * sizeof(uint32_t), stream) != hipSuccess) return MI355GS_ELAUNCH;
    t->consts_ready = true;
  }
  struct HookScope {
    HookScope(float* gate, const GsPrologue& pro, const GsPosed& posed, int min_units, int det) {
      gs_pin_min_units(min_units);
      gs_pin_deterministic(det);
      g_fused.skip_memsets = true; g_fused.gate = gate; g_fused.prologue = pro; g_fused.posed = posed;
      g_fused.gate_xyz = 0; g_fused.gate_sh = 1; g_fused.gate_sh_rest = 2; g_fused.gate_opacity = 3; g_fused.gate_scaling = 4; g_fused.gate_rot = 5; g_fused.gate_pose = 6;
    }
    ~HookScope() { g_fused = GsFusedStepHooks(); gs_pin_min_units(0); gs_pin_deterministic(-1); }
  };
  GsPrologue pro;  // the step's accumulators are cleared by its first kernel (k_pose_fwd)
  {
    const TilesLayout tl(W, H);
    pro.grad_records = (float4*)t->grad_scratch; pro.n_vec = (size_t)P * 3;
    pro.tile_counters = (uint32_t*)(t->tiles + tl.count); pro.n_counters = (int)((tl.start - tl.count) / 4);
    pro.g_poses = t->g_poses; pro.n_pose = 7 * t->V; pro.pose_scratch = t->pose_scratch; pro.adam_scratch = t->adam_scratch;
  }
  GsPosed posed;
  posed.pose = t->poses + 7 * (size_t)view; posed.acc = t->pose_scratch; posed.partial = t->pose_partial;
  HookScope hook_scope(t->adam_scratch, pro, posed, t->min_units, t->det);
  const float* view_m = t->consts;
  const float* campos = t->consts + 16;
  const float* pose = t->poses + 7 * (size_t)view;
  int rc;
  // ---- forward: the projection kernel applies the camera-frame transform and the activations itself (GsPosed)
  // degree 0 reads only the DC coefficient; higher degrees read f_dc + f_rest in place (split storage)
  const int D = sh_degree, M = D == 0 ? 1 : 16;
  const float* rest = D == 0 ? nullptr : t->f_rest;
  float* g_rest = D == 0 ? nullptr : t->g_frest;
  if ((rc = mi355gs_raster_forward_preprocess(stream, P, D, M, W, H, t->xyz, t->f_dc, rest, nullptr, t->opacity, t->scaling, 1.0f,
                                              t->rotation, nullptr, view_m, projmatrix, campos, tanfovx, tanfovy, 0, t->radii,
                                              t->geom, t->tiles, num_rendered_out, nullptr, nullptr, 0)))
    return rc;
  if ((rc = mi355gs_raster_forward_render(stream, P, W, H, t->capacity, bg, t->geom, t->tiles, t->binning, t->image, 0))) return rc;
  // ---- loss and its gradient in one launch (the SSIM partial-derivative maps never leave LDS); the loss VALUE, which only
  // the host ever reads, is summed from the per-workgroup partials by the single-workgroup pose-finish kernel further down
  if ((rc = gs_loss_fused(stream, 3, H, W, t->image, gt_image, lambda_dssim, t->dL_dimg, t->ssim_scratch))) return rc;
  // ---- backward: down to the raw-parameter gradients and the pose sums in one kernel, then the 7 pose gradients
  if ((rc = mi355gs_raster_backward(stream, P, D, M, W, H, bg, t->xyz, t->f_dc, rest, nullptr, t->opacity, t->scaling, 1.0f, t->rotation,
                                    nullptr, view_m, projmatrix, campos, tanfovx, tanfovy, t->geom, t->tiles, t->binning,
                                    t->capacity, t->radii, t->image, t->dL_dimg, t->grad_scratch, t->g_xyz, t->g_means2D, t->g_fdc,
                                    g_rest, t->g_colors, t->g_opacity, t->g_scaling, t->g_rot, nullptr, 0, 0)))
    return rc;
  GS_KRANGE("pose_finish");
  gs_launch_pose_finish_partials(stream, pose, t->pose_partial, (P + 255) / 256, t->g_poses + 7 * (size_t)view, t->adam_scratch + 6,
                                 (const float*)t->ssim_scratch, gs_loss_fused_nblocks(3, H, W), 1.0 / (3.0 * H * W), lambda_dssim, loss_out, 0, 0);
  GS_CHECK_LAUNCH("pose_finish");
  // ---- optimizer: groups in the reference's order xyz, f_dc, f_rest, opacity, scaling, rotation, pose
  if (do_optimizer_step) return trainer_adam(t, stream, lr, step, beta1, beta2, eps, true);
  return MI355GS_OK;
}

int mi355gs_trainer_optimizer_step(void* handle, void* stream_, const float* lr, const int32_t* step, float beta1, float beta2,
                                   float eps, int commit_gate) {
  GS_RANGE();
  Trainer* t = (Trainer*)handle;
  if (!t || !lr || !step || !t->consts_ready) return MI355GS_EINVAL;  // needs the gradients of a preceding step
  // gate flags of the gradients produced by the preceding mi355gs_trainer_step(..., do_optimizer_step = 0) are still in place
  g_fused.gate = t->adam_scratch;
  // commit_gate = 0: the caller has seen the count; 1: it has not — the launch decides on the device like a one-call step's
  const int rc = trainer_adam(t, (hipStream_t)stream_, lr, step, beta1, beta2, eps, commit_gate != 0);
  g_fused = GsFusedStepHooks();
  return rc;
}

}  // extern "C"
