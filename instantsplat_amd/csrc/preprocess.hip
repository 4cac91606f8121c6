// Per-Gaussian stages of the rasterizer: projection (forward) and its closed-form backward.
// One thread per Gaussian, 256-thread blocks; HBM-bound (131 B/Gaussian at SH
// degree 0, 311 B at degree 3 forward).  Semantics: SURVEY.md Appendix A.1 / A.5; the operator
// being replaced is reached at reference gaussian_renderer/__init__.py:126-135.
#include "pose_math.h"

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

struct Cov2D {
  float a, b, c;  // after the +0.3 low-pass
};

// Shared by forward and backward so both see bit-identical intermediates.
struct Ewa {
  float tx, ty, tz, xmul, ymul;
  float M[6];  // J * W (2x3)
};

__device__ __forceinline__ Ewa ewa_setup(float3 pv, const float* view, const CamParams& cp) {
  Ewa e;
  const float limx = 1.3f * cp.tanfovx, limy = 1.3f * cp.tanfovy;
  const float txtz = pv.x / pv.z, tytz = pv.y / pv.z;
  e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv.z;
  e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv.z;
  e.tz = pv.z;
  e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float j00 = cp.focal_x / e.tz, j02 = -(cp.focal_x * e.tx) / (e.tz * e.tz);
  const float j11 = cp.focal_y / e.tz, j12 = -(cp.focal_y * e.ty) / (e.tz * e.tz);
  // W rows: (view0, view4, view8), (view1, view5, view9), (view2, view6, view10)
  e.M[0] = j00 * view[0] + j02 * view[2];
  e.M[1] = j00 * view[4] + j02 * view[6];
  e.M[2] = j00 * view[8] + j02 * view[10];
  e.M[3] = j11 * view[1] + j12 * view[2];
  e.M[4] = j11 * view[5] + j12 * view[6];
  e.M[5] = j11 * view[9] + j12 * view[10];
  return e;
}

__device__ __forceinline__ Cov2D ewa_cov2d(const Ewa& e, const float* cov) {
  // MS = M * Sigma (2x3), cov2D = MS * M^T
  const float* M = e.M;
  float MS[6];
  MS[0] = M[0] * cov[0] + M[1] * cov[1] + M[2] * cov[2];
  MS[1] = M[0] * cov[1] + M[1] * cov[3] + M[2] * cov[4];
  MS[2] = M[0] * cov[2] + M[1] * cov[4] + M[2] * cov[5];
  MS[3] = M[3] * cov[0] + M[4] * cov[1] + M[5] * cov[2];
  MS[4] = M[3] * cov[1] + M[4] * cov[3] + M[5] * cov[4];
  MS[5] = M[3] * cov[2] + M[4] * cov[4] + M[5] * cov[5];
  Cov2D c;
  c.a = MS[0] * M[0] + MS[1] * M[1] + MS[2] * M[2] + 0.3f;
  c.b = MS[0] * M[3] + MS[1] * M[4] + MS[2] * M[5];
  c.c = MS[3] * M[3] + MS[4] * M[4] + MS[5] * M[5] + 0.3f;
  return c;
}

__device__ __forceinline__ void quat_to_R(float r, float x, float y, float z, float* R) {
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// SH basis for unit direction d, order/signs of reference utils/sh_utils.py:74-100
template <int NB>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
  b[0] = SH_C0;
  if (NB > 1) { b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x; }
  if (NB > 4) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy);
    b[7] = SH_C2_3 * xz; b[8] = SH_C2_4 * (xx - yy);
    if (NB > 9) {
      b[9] = SH_C3_0 * y * (3.f * xx - yy); b[10] = SH_C3_1 * xy * z;
      b[11] = SH_C3_2 * y * (4.f * zz - xx - yy); b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
      b[13] = SH_C3_4 * x * (4.f * zz - xx - yy); b[14] = SH_C3_5 * z * (xx - yy);
      b[15] = SH_C3_6 * x * (xx - 3.f * yy);
    }
  }
}

template <int NB>
__device__ __forceinline__ float3 sh_eval(const float* sh /*[M,3]*/, float x, float y, float z) {
  float b[NB];
  sh_basis<NB>(x, y, z, b);
  float3 acc = make_float3(0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    acc.x += b[k] * sh[3 * k]; acc.y += b[k] * sh[3 * k + 1]; acc.z += b[k] * sh[3 * k + 2];
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// K1 forward: projection + per-tile instance counting
// ------------------------------------------------------------------------------------------------
// POSED (one-call train step, see GsPosed): means3D / rotations / scales / opacities are the RAW parameters (xyz, raw
// quaternion, log-scale, opacity logit); the camera-frame transform and the activations are applied here, and the step's
// accumulators are cleared on the way (this is then the first kernel of the step).
template <bool POSED, int D>
__global__ __launch_bounds__(256) void k_preprocess_fwd(
    int P, int M, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
    const float* __restrict__ colors_precomp, const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, CamParams cp,
    int32_t* __restrict__ radii, GsRec* __restrict__ recs, float* __restrict__ cov3Ds, uint2* __restrict__ rects,
    uint8_t* __restrict__ clamped, float* __restrict__ depths, uint8_t* __restrict__ visible, GsPosed posed, GsPrologue pro) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // The frame's accumulators are cleared by its first kernel — this one — instead of memsets in front of it (a 2-4 us launch
  // and a dependent-dispatch boundary each): the per-tile counters always; the backward's moment records (and the gate flags
  // behind them) when the caller handed them over already; the one-call step's pose / optimizer scratch.
  {
    const size_t gid = (size_t)i, stride = (size_t)gridDim.x * blockDim.x;
    if (pro.grad_records) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (size_t k = gid; k < pro.n_vec; k += stride) pro.grad_records[k] = z;
    }
    if (pro.tile_counters)
      for (size_t k = gid; k < (size_t)pro.n_counters; k += stride) pro.tile_counters[k] = 0u;
    if (pro.g_poses && gid < (size_t)pro.n_pose) pro.g_poses[gid] = 0.f;
    if (pro.pose_scratch && gid < 32) pro.pose_scratch[gid] = 0.f;
    if (pro.adam_scratch && gid < 8) pro.adam_scratch[gid] = 0.f;
  }
  if (i >= P) return;
  radii[i] = 0;
  if (visible) visible[i] = 0;
  rects[i] = make_uint2(0u, 0u);
  const float* view = cp.view;
  const float* proj = cp.proj;
  PoseMat pm;
  if (POSED) pm = load_pose(posed.pose);
  float3 m = make_float3(means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]);
  if (POSED) m = pose_mean(pm, m.x, m.y, m.z);
  const float3 pv = gs_tp43(view, m);
  if (pv.z <= 0.2f) return;
  const float4 ph = gs_tp44(proj, m);
  const float pw = 1.0f / (ph.w + 0.0000001f);
  const float ndcx = ph.x * pw, ndcy = ph.y * pw;

  float cov[6];
  if (cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = cov3D_precomp[6 * (size_t)i + k];
  } else {
    float4 q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
    float sc[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
    if (POSED) {
      q = pose_rot(pm, q);
#pragma unroll
      for (int k = 0; k < 3; ++k) sc[k] = pose_scale(sc[k]);
    }
    float R[9];
    quat_to_R(q.x, q.y, q.z, q.w, R);
    const float s0 = cp.scale_modifier * sc[0], s1 = cp.scale_modifier * sc[1], s2 = cp.scale_modifier * sc[2];
    float L[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) { L[3 * r] = R[3 * r] * s0; L[3 * r + 1] = R[3 * r + 1] * s1; L[3 * r + 2] = R[3 * r + 2] * s2; }
    cov[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    cov[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    cov[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    cov[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    cov[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    cov[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) cov3Ds[6 * (size_t)i + k] = cov[k];

  const Ewa e = ewa_setup(pv, view, cp);
  const Cov2D c2 = ewa_cov2d(e, cov);
  const float det = c2.a * c2.c - c2.b * c2.b;
  if (det == 0.0f) return;
  const float det_inv = 1.f / det;
  const float ca = c2.c * det_inv, cb = -c2.b * det_inv, cc = c2.a * det_inv;
  const float mid = 0.5f * (c2.a + c2.c);
  const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
  const float radius = ceilf(3.f * sqrtf(fmaxf(mid + sq, mid - sq)));
  const float px = ((ndcx + 1.0f) * cp.W - 1.0f) * 0.5f, py = ((ndcy + 1.0f) * cp.H - 1.0f) * 0.5f;
  const int rminx = min(cp.gx, max(0, (int)((px - radius) / GS_TILE)));
  const int rminy = min(cp.gy, max(0, (int)((py - radius) / GS_TILE)));
  const int rmaxx = min(cp.gx, max(0, (int)((px + radius + GS_TILE - 1) / GS_TILE)));
  const int rmaxy = min(cp.gy, max(0, (int)((py + radius + GS_TILE - 1) / GS_TILE)));
  if ((rmaxx - rminx) * (rmaxy - rminy) == 0) return;

  float3 rgb;
  uint8_t clamp_bits = 0;
  if (colors_precomp) {
    rgb = make_float3(colors_precomp[3 * (size_t)i], colors_precomp[3 * (size_t)i + 1], colors_precomp[3 * (size_t)i + 2]);
  } else {
    float dx = m.x - cp.campos[0], dy = m.y - cp.campos[1], dz = m.z - cp.campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
    float3 v;
    constexpr int NBF = (D + 1) * (D + 1);
    if (shs_rest == nullptr) {
      v = sh_eval<NBF>(shs + (size_t)i * M * 3, dx, dy, dz);
    } else {
      // split storage (the reference's own parameter layout: _features_dc [P,1,3] + _features_rest [P,M-1,3]):
      // no cat(f_dc, f_rest) has to be materialised for the rasterizer
      float shl[NBF * 3];
      shl[0] = shs[3 * (size_t)i]; shl[1] = shs[3 * (size_t)i + 1]; shl[2] = shs[3 * (size_t)i + 2];
      const float* rest = shs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
      for (int k = 3; k < NBF * 3; ++k) shl[k] = rest[k - 3];
      v = sh_eval<NBF>(shl, dx, dy, dz);
    }
    v.x += 0.5f; v.y += 0.5f; v.z += 0.5f;
    clamp_bits = (uint8_t)((v.x < 0.f ? 1 : 0) | (v.y < 0.f ? 2 : 0) | (v.z < 0.f ? 4 : 0));
    rgb = make_float3(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f));
  }

  // Half extents of the axis-aligned box around {alpha >= 1/255}: a pixel outside it would be skipped
  // by the alpha test anyway, so the composite kernels use the box to skip whole 8x8 pixel groups.
  // Computed from the conic actually used; 1% + 0.5 px slack; disabled (inf) when the conic is too
  // ill-conditioned for the bound to be trusted.
  const float opac = POSED ? pose_opacity(opacities[i]) : opacities[i];
  float hx, hy;
  const float tau = __logf(255.0f * opac);
  const float cdet = ca * cc - cb * cb;
  if (!(tau > 0.f)) { hx = hy = -1e30f; }
  else if (!(cdet > 0.f) || ca * cc > 1000.f * cdet) { hx = hy = 1e30f; }
  else {
    const float s = 2.f * tau / cdet;
    hx = sqrtf(s * cc) * 1.01f + 0.5f;
    hy = sqrtf(s * ca) * 1.01f + 0.5f;
  }

  radii[i] = (int)radius;
  if (visible) visible[i] = 1;   // the reference's `visibility_filter` (radii > 0; radius >= 1 here) without a compare kernel behind the operator
  clamped[i] = clamp_bits;
  // Tiles this Gaussian is binned to: the reference's 3-sigma rect, intersected with the tiles the
  // alpha >= 1/255 box can reach.  Dropped tiles hold no pixel that would pass the alpha test, so the
  // image, radii and gradients are unchanged while the instance count R shrinks.
  int bminx = rminx, bminy = rminy, bmaxx = rmaxx, bmaxy = rmaxy;
  if (hx < 1e29f) {
    if (hx < 0.f) { bmaxx = bminx; bmaxy = bminy; }
    else {
      bminx = max(bminx, (int)floorf((px - hx) * (1.0f / GS_TILE)));
      bminy = max(bminy, (int)floorf((py - hy) * (1.0f / GS_TILE)));
      bmaxx = min(bmaxx, (int)floorf((px + hx) * (1.0f / GS_TILE)) + 1);
      bmaxy = min(bmaxy, (int)floorf((py + hy) * (1.0f / GS_TILE)) + 1);
      if (bmaxx <= bminx || bmaxy <= bminy) { bmaxx = bminx; bmaxy = bminy; }
    }
  }
  rects[i] = make_uint2((uint32_t)bminx | ((uint32_t)bminy << 16), (uint32_t)bmaxx | ((uint32_t)bmaxy << 16));
  GsRec rec;
  rec.q0 = make_float4(px, py, hx, hy);
  rec.q1 = make_float4(-0.5f * GS_LOG2E * ca, -0.5f * GS_LOG2E * cc, -GS_LOG2E * cb, opac);
  rec.q2 = make_float4(rgb.x, rgb.y, rgb.z, pv.z);
  recs[i] = rec;
  depths[i] = pv.z;
}

// ------------------------------------------------------------------------------------------------
// K8+K9 backward: screen-space gradients -> means, SH, opacity, scale, rotation (A.5)
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void sh_backward(const float* sh, float* gsh, float x, float y, float z, float3 gr, float* gd) {
  float b[NB];
  sh_basis<NB>(x, y, z, b);
#pragma unroll
  for (int k = 0; k < NB; ++k) { gsh[3 * k] = b[k] * gr.x; gsh[3 * k + 1] = b[k] * gr.y; gsh[3 * k + 2] = b[k] * gr.z; }
  // w_k = sum_c sh[k,c] * g_c ; gd = sum_k dbasis_k/dd * w_k
  float w[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) w[k] = sh[3 * k] * gr.x + sh[3 * k + 1] * gr.y + sh[3 * k + 2] * gr.z;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (NB > 1) { gy += -SH_C1 * w[1]; gz += SH_C1 * w[2]; gx += -SH_C1 * w[3]; }
  if (NB > 4) {
    const float xx = x * x, yy = y * y, zz = z * z;
    gx += SH_C2_0 * y * w[4]; gy += SH_C2_0 * x * w[4];
    gy += SH_C2_1 * z * w[5]; gz += SH_C2_1 * y * w[5];
    gx += SH_C2_2 * -2.f * x * w[6]; gy += SH_C2_2 * -2.f * y * w[6]; gz += SH_C2_2 * 4.f * z * w[6];
    gx += SH_C2_3 * z * w[7]; gz += SH_C2_3 * x * w[7];
    gx += SH_C2_4 * 2.f * x * w[8]; gy += SH_C2_4 * -2.f * y * w[8];
    if (NB > 9) {
      gx += SH_C3_0 * 6.f * x * y * w[9]; gy += SH_C3_0 * (3.f * xx - 3.f * yy) * w[9];
      gx += SH_C3_1 * y * z * w[10]; gy += SH_C3_1 * x * z * w[10]; gz += SH_C3_1 * x * y * w[10];
      gx += SH_C3_2 * -2.f * x * y * w[11]; gy += SH_C3_2 * (4.f * zz - xx - 3.f * yy) * w[11]; gz += SH_C3_2 * 8.f * y * z * w[11];
      gx += SH_C3_3 * -6.f * x * z * w[12]; gy += SH_C3_3 * -6.f * y * z * w[12]; gz += SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * w[12];
      gx += SH_C3_4 * (4.f * zz - 3.f * xx - yy) * w[13]; gy += SH_C3_4 * -2.f * x * y * w[13]; gz += SH_C3_4 * 8.f * x * z * w[13];
      gx += SH_C3_5 * 2.f * x * z * w[14]; gy += SH_C3_5 * -2.f * y * z * w[14]; gz += SH_C3_5 * (xx - yy) * w[14];
      gx += SH_C3_6 * (3.f * xx - 3.f * yy) * w[15]; gy += SH_C3_6 * -6.f * x * y * w[15];
    }
  }
  gd[0] = gx; gd[1] = gy; gd[2] = gz;
}

// POSED: see k_preprocess_fwd; the outputs named dL_dmeans3D / dL_drots / dL_dscales / dL_dopac then receive the gradients
// of the RAW parameters (xyz, raw quaternion, log-scale, opacity logit) and the 16 pose sums are accumulated.
template <bool POSED, int D>
__global__ __launch_bounds__(256) void k_preprocess_bwd(
    int P, int M, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ shs_rest,
    const float* __restrict__ scales, const float* __restrict__ rotations, int use_shs, int use_cov_precomp, CamParams cp, const int32_t* __restrict__ radii,
    const GsRec* __restrict__ recs, const float* __restrict__ cov3Ds, const uint8_t* __restrict__ clamped, const GsGrad* __restrict__ grads,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dshs_rest,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopac, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, float* __restrict__ sh_gate, float* __restrict__ sh_rest_gate,
    GsPosed posed, float* __restrict__ gate, int gi_xyz, int gi_rot, int gi_scaling, int gi_opacity) {
  __shared__ float s_red[POSED ? 4 : 1][16];
  const int i_raw = blockIdx.x * blockDim.x + threadIdx.x;
  if (!POSED && i_raw >= P) return;
  const bool live = i_raw < P;          // POSED: every thread stays for the workgroup reduction of the pose sums
  const int i = live ? i_raw : 0;
  const bool vis = live && radii[i] > 0;
  PoseMat pm;
  if (POSED) pm = load_pose(posed.pose);
  GsGrad g;
  g.g0 = make_float4(0, 0, 0, 0); g.g1 = g.g0; g.g2 = g.g0;
  if (vis) {
    // moments (see GsGrad) -> screen-space gradients, with this Gaussian's conic (a, b, c) and opacity
    const GsGrad mo = grads[i];
    const float4 pre = recs[i].q1;  // undo the exp2 pre-scaling and ordering: back to the conic (a, b, c)
    const float4 con = make_float4(pre.x * (-2.0f / GS_LOG2E), pre.z * (-1.0f / GS_LOG2E), pre.y * (-2.0f / GS_LOG2E), pre.w);
    const float m1 = mo.g0.x, m2 = mo.g0.y, m3 = mo.g0.z, m4 = mo.g0.w, m5 = mo.g1.x, m0 = mo.g1.y;
    g.g0.x = -(con.x * m1 + con.y * m2) * (0.5f * cp.W);   // dL/dmean2D.x (NDC-scaled, as the reference reports it)
    g.g0.y = -(con.z * m2 + con.y * m1) * (0.5f * cp.H);   // dL/dmean2D.y
    g.g0.z = -0.5f * m3;                                    // dL/dconic a
    g.g0.w = -m4;                                           // dL/dconic b (both off-diagonal entries)
    g.g1.x = -0.5f * m5;                                    // dL/dconic c
    g.g1.y = con.w != 0.f ? m0 / con.w : 0.f;               // dL/dopacity
    g.g1.z = mo.g1.z; g.g1.w = mo.g1.w; g.g2.x = mo.g2.x;   // dL/drgb
  }
  float gm[3] = {0.f, 0.f, 0.f};
  float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  const float3 gcol = make_float3(g.g1.z, g.g1.w, g.g2.x);
  constexpr int nb = (D + 1) * (D + 1);
  const bool split = shs_rest != nullptr;
  float gsh_local[nb * 3];  // SH gradient of the active bands, in registers; stored (and zero-padded) at the end

  if (vis) {
    const float* view = cp.view;
    const float* proj = cp.proj;
    float3 m = make_float3(means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]);
    if (POSED) m = pose_mean(pm, m.x, m.y, m.z);
    float cov[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = cov3Ds[6 * (size_t)i + k];
    // ---- conic -> cov2D -> (cov3D, mean)
    {
      const float3 pv = gs_tp43(view, m);
      const Ewa e = ewa_setup(pv, view, cp);
      const Cov2D c2 = ewa_cov2d(e, cov);
      const float a = c2.a, b = c2.b, c = c2.c;
      const float det = a * c - b * b;
      const float Dv = 1.f / (det * det + 0.0000001f);
      const float gA = g.g0.z, gB = g.g0.w, gC = g.g1.x;
      const float ga = Dv * (-c * c * gA + b * c * gB + (det - a * c) * gC);
      const float gc = Dv * ((det - a * c) * gA + a * b * gB - a * a * gC);
      const float gb = Dv * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
      const float* Mx = e.M;
      float GM[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) { GM[k] = ga * Mx[k] + 0.5f * gb * Mx[3 + k]; GM[3 + k] = 0.5f * gb * Mx[k] + gc * Mx[3 + k]; }
      // gSigma = M^T Gm M, packed with doubled off-diagonals
      gcov[0] = Mx[0] * GM[0] + Mx[3] * GM[3];
      gcov[1] = 2.f * (Mx[0] * GM[1] + Mx[3] * GM[4]);
      gcov[2] = 2.f * (Mx[0] * GM[2] + Mx[3] * GM[5]);
      gcov[3] = Mx[1] * GM[1] + Mx[4] * GM[4];
      gcov[4] = 2.f * (Mx[1] * GM[2] + Mx[4] * GM[5]);
      gcov[5] = Mx[2] * GM[2] + Mx[5] * GM[5];
      // gM = 2 * GM * Sigma
      float gM[6];
      gM[0] = 2.f * (GM[0] * cov[0] + GM[1] * cov[1] + GM[2] * cov[2]);
      gM[1] = 2.f * (GM[0] * cov[1] + GM[1] * cov[3] + GM[2] * cov[4]);
      gM[2] = 2.f * (GM[0] * cov[2] + GM[1] * cov[4] + GM[2] * cov[5]);
      gM[3] = 2.f * (GM[3] * cov[0] + GM[4] * cov[1] + GM[5] * cov[2]);
      gM[4] = 2.f * (GM[3] * cov[1] + GM[4] * cov[3] + GM[5] * cov[4]);
      gM[5] = 2.f * (GM[3] * cov[2] + GM[4] * cov[4] + GM[5] * cov[5]);
      // gJ = gM * W^T ; only J00, J02, J11, J12 vary
      const float gJ00 = gM[0] * view[0] + gM[1] * view[4] + gM[2] * view[8];
      const float gJ02 = gM[0] * view[2] + gM[1] * view[6] + gM[2] * view[10];
      const float gJ11 = gM[3] * view[1] + gM[4] * view[5] + gM[5] * view[9];
      const float gJ12 = gM[3] * view[2] + gM[4] * view[6] + gM[5] * view[10];
      const float tz2 = 1.f / (e.tz * e.tz), tz3 = tz2 / e.tz;
      const float gtx = e.xmul * -cp.focal_x * tz2 * gJ02;
      const float gty = e.ymul * -cp.focal_y * tz2 * gJ12;
      const float gtz = -cp.focal_x * tz2 * gJ00 - cp.focal_y * tz2 * gJ11 + 2.f * cp.focal_x * e.tx * tz3 * gJ02 +
                        2.f * cp.focal_y * e.ty * tz3 * gJ12;
      gm[0] = view[0] * gtx + view[1] * gty + view[2] * gtz;
      gm[1] = view[4] * gtx + view[5] * gty + view[6] * gtz;
      gm[2] = view[8] * gtx + view[9] * gty + view[10] * gtz;
    }
    // ---- projection path (screen-space mean gradient)
    {
      const float4 ph = gs_tp44(proj, m);
      const float mw = 1.0f / (ph.w + 0.0000001f);
      const float mul1 = ph.x * mw * mw, mul2 = ph.y * mw * mw;
      const float g2x = g.g0.x, g2y = g.g0.y;
      gm[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
      gm[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
      gm[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
    }
    // ---- SH
    if (use_shs) {
      float dx = m.x - cp.campos[0], dy = m.y - cp.campos[1], dz = m.z - cp.campos[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      dx *= inv; dy *= inv; dz *= inv;
      const uint8_t cl = clamped[i];
      const float3 gr = make_float3((cl & 1) ? 0.f : gcol.x, (cl & 2) ? 0.f : gcol.y, (cl & 4) ? 0.f : gcol.z);
      float shl[nb * 3];  // the active coefficients, in registers (from one [P,M,3] tensor or from f_dc + f_rest)
      if (split) {
        shl[0] = shs[3 * (size_t)i]; shl[1] = shs[3 * (size_t)i + 1]; shl[2] = shs[3 * (size_t)i + 2];
        const float* rest = shs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
        for (int k = 3; k < nb * 3; ++k) shl[k] = rest[k - 3];
      } else {
        const float* src = shs + (size_t)i * M * 3;
#pragma unroll
        for (int k = 0; k < nb * 3; ++k) shl[k] = src[k];
      }
      float gd[3];
      sh_backward<nb>(shl, gsh_local, dx, dy, dz, gr, gd);
      const float dot = dx * gd[0] + dy * gd[1] + dz * gd[2];
      gm[0] += (gd[0] - dx * dot) * inv; gm[1] += (gd[1] - dy * dot) * inv; gm[2] += (gd[2] - dz * dot) * inv;
    }
    // ---- cov3D -> scale, rotation
    if (!use_cov_precomp) {
      float4 q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
      float sc[3] = {scales[3 * (size_t)i], scales[3 * (size_t)i + 1], scales[3 * (size_t)i + 2]};
      if (POSED) {
        q = pose_rot(pm, q);
#pragma unroll
        for (int k = 0; k < 3; ++k) sc[k] = pose_scale(sc[k]);
      }
      float R[9];
      quat_to_R(q.x, q.y, q.z, q.w, R);
      const float mod = cp.scale_modifier;
      const float s[3] = {mod * sc[0], mod * sc[1], mod * sc[2]};
      float L[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) { L[3 * r] = R[3 * r] * s[0]; L[3 * r + 1] = R[3 * r + 1] * s[1]; L[3 * r + 2] = R[3 * r + 2] * s[2]; }
      const float Gs[9] = {gcov[0], 0.5f * gcov[1], 0.5f * gcov[2], 0.5f * gcov[1], gcov[3], 0.5f * gcov[4],
                           0.5f * gcov[2], 0.5f * gcov[4], gcov[5]};
      float gL[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) gL[3 * r + k] = 2.f * (Gs[3 * r] * L[k] + Gs[3 * r + 1] * L[3 + k] + Gs[3 * r + 2] * L[6 + k]);
      float Rp[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // the gradient with respect to the MODIFIED scale, times cp.scale_grad_factor: 1 by default — the published operator's
        // computeCov3D backward forms s = mod * scale first and returns dL/ds as dL/dscale — or mod (the true derivative) after
        // mi355gs_tune_scale_grad(1).  The same number at mod = 1, the only value the reference trains with.
        gs[k] = cp.scale_grad_factor * (R[k] * gL[k] + R[3 + k] * gL[3 + k] + R[6 + k] * gL[6 + k]);
#pragma unroll
        for (int r = 0; r < 3; ++r) Rp[3 * r + k] = gL[3 * r + k] * s[k];
      }
      const float r_ = q.x, x = q.y, y = q.z, z = q.w;
      gq[0] = 2.f * (z * (Rp[3] - Rp[1]) + y * (Rp[2] - Rp[6]) + x * (Rp[7] - Rp[5]));
      gq[1] = 2.f * (y * (Rp[1] + Rp[3]) + z * (Rp[2] + Rp[6]) + r_ * (Rp[7] - Rp[5])) - 4.f * x * (Rp[4] + Rp[8]);
      gq[2] = 2.f * (x * (Rp[1] + Rp[3]) + r_ * (Rp[2] - Rp[6]) + z * (Rp[5] + Rp[7])) - 4.f * y * (Rp[0] + Rp[8]);
      gq[3] = 2.f * (r_ * (Rp[3] - Rp[1]) + x * (Rp[2] + Rp[6]) + y * (Rp[5] + Rp[7])) - 4.f * z * (Rp[0] + Rp[4]);
    }
  }

  // ---- write every output row (zeros for culled Gaussians: callers get fully-defined tensors)
  if (live) {
    if (!POSED) {
      dL_dmeans3D[3 * (size_t)i] = gm[0]; dL_dmeans3D[3 * (size_t)i + 1] = gm[1]; dL_dmeans3D[3 * (size_t)i + 2] = gm[2];
      dL_dopac[i] = g.g1.y;
      if (dL_dscales) { dL_dscales[3 * (size_t)i] = gs[0]; dL_dscales[3 * (size_t)i + 1] = gs[1]; dL_dscales[3 * (size_t)i + 2] = gs[2]; }
      if (dL_drots) *reinterpret_cast<float4*>(dL_drots + 4 * (size_t)i) = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
    dL_dmeans2D[3 * (size_t)i] = g.g0.x; dL_dmeans2D[3 * (size_t)i + 1] = g.g0.y; dL_dmeans2D[3 * (size_t)i + 2] = 0.f;
    if (dL_dcolors) { dL_dcolors[3 * (size_t)i] = gcol.x; dL_dcolors[3 * (size_t)i + 1] = gcol.y; dL_dcolors[3 * (size_t)i + 2] = gcol.z; }
    if (dL_dshs) {
      const bool have = vis && use_shs;  // gsh_local was written by sh_backward
      bool nz_dc = false, nz_rest = false;
      if (!split) {
        float* gdst = dL_dshs + (size_t)i * M * 3;
#pragma unroll
        for (int k = 0; k < nb * 3; ++k) { const float v = have ? gsh_local[k] : 0.f; gdst[k] = v; nz_dc = nz_dc || v != 0.f; }
        for (int k = nb * 3; k < M * 3; ++k) gdst[k] = 0.f;  // bands above the active degree
      } else {
        // the two parameter tensors of the split storage; untouched coefficients get explicit zeros
        float* gdc = dL_dshs + 3 * (size_t)i;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = have ? gsh_local[c] : 0.f; gdc[c] = v; nz_dc = nz_dc || v != 0.f; }
        if (dL_dshs_rest) {
          float* grest = dL_dshs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
          for (int k = 3; k < nb * 3; ++k) { const float v = have ? gsh_local[k] : 0.f; grest[k - 3] = v; nz_rest = nz_rest || v != 0.f; }
          for (int k = nb * 3; k < M * 3; ++k) grest[k - 3] = 0.f;  // bands above the active degree
        }
      }
      // Adam's whole-tensor gate flags (benign race: every writer stores the same value)
      if (sh_gate && nz_dc) *sh_gate = 1.0f;
      if (sh_rest_gate && nz_rest) *sh_rest_gate = 1.0f;
    }
    if (dL_dcov3D) {
      const bool on = vis && use_cov_precomp;
#pragma unroll
      for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] = on ? gcov[k] : 0.f;
    }
  }
  if (POSED) {
    // camera-frame gradients -> raw-parameter gradients and the pose sums (what k_pose_bwd does on the autograd path)
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.f;
    bool nz_xyz = false, nz_rot = false, nz_sc = false, nz_op = false;
    if (live) {
      float act[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) act[k] = pose_scale(scales[3 * (size_t)i + k]);
      const float o = vis ? recs[i].q1.w : 0.5f;  // the activated opacity of the forward (its gradient is zero when culled)
      const PoseGradOut r = pose_backward_one(pm, means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2],
                                              *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i), act, o, gm[0], gm[1], gm[2],
                                              make_float4(gq[0], gq[1], gq[2], gq[3]), gs, g.g1.y, a);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dL_dmeans3D[3 * (size_t)i + k] = r.d_xyz[k];
        dL_dscales[3 * (size_t)i + k] = r.d_scaling[k];
        nz_xyz = nz_xyz || r.d_xyz[k] != 0.f;
        nz_sc = nz_sc || r.d_scaling[k] != 0.f;
      }
      *reinterpret_cast<float4*>(dL_drots + 4 * (size_t)i) = r.d_rot;
      nz_rot = nz_rot || r.d_rot.x != 0.f || r.d_rot.y != 0.f || r.d_rot.z != 0.f || r.d_rot.w != 0.f;
      dL_dopac[i] = r.d_opacity_logit;
      nz_op = nz_op || r.d_opacity_logit != 0.f;
    }
    pose_accumulate(a, posed.acc, posed.partial, s_red);
    if (gate) {  // PerPointAdam's whole-tensor gate: any non-zero gradient element (benign same-value store race)
      if (gi_xyz >= 0 && nz_xyz) gate[gi_xyz] = 1.0f;
      if (gi_rot >= 0 && nz_rot) gate[gi_rot] = 1.0f;
      if (gi_scaling >= 0 && nz_sc) gate[gi_scaling] = 1.0f;
      if (gi_opacity >= 0 && nz_op) gate[gi_opacity] = 1.0f;
    }
  }
}

__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                                      uint8_t* __restrict__ present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float3 m = make_float3(means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]);
  present[i] = gs_tp43(view, m).z > 0.2f ? 1 : 0;
}

}  // namespace

// ---- host-side launchers (called from api.hip)
int gs_launch_preprocess_fwd(hipStream_t stream, int P, int D, int M, const float* means3D, const float* shs, const float* shs_rest,
                             const float* colors_precomp, const float* opacities, const float* scales,
                             const float* rotations, const float* cov3D_precomp, const CamParams& cp, int32_t* radii,
                             GsRec* recs, float* cov3Ds, uint2* rects, uint8_t* clamped, float* depths, uint8_t* visible,
                             const GsPrologue& pro) {
  if (P <= 0) return 0;
  const bool posed = g_fused.posed.pose != nullptr;
  const GsPosed pa = posed ? g_fused.posed : GsPosed();
#define GS_FWD(POSED, DEG)                                                                                                              \
  hipLaunchKernelGGL((k_preprocess_fwd<POSED, DEG>), dim3((P + 255) / 256), dim3(256), 0, stream, P, M, means3D, shs, shs_rest,         \
                     colors_precomp, opacities, scales, rotations, cov3D_precomp, cp, radii, recs, cov3Ds, rects, clamped, depths, visible, pa, pro)
  const int deg = shs ? D : 0;  // one instantiation per active SH degree: coefficient arrays stay in registers
  if (posed) { if (deg == 0) GS_FWD(true, 0); else if (deg == 1) GS_FWD(true, 1); else if (deg == 2) GS_FWD(true, 2); else GS_FWD(true, 3); }
  else { if (deg == 0) GS_FWD(false, 0); else if (deg == 1) GS_FWD(false, 1); else if (deg == 2) GS_FWD(false, 2); else GS_FWD(false, 3); }
#undef GS_FWD
  return 0;
}

int gs_launch_preprocess_bwd(hipStream_t stream, int P, int D, int M, const float* means3D, const float* shs, const float* shs_rest,
                             const float* scales, const float* rotations, int use_shs, int use_cov_precomp,
                             const CamParams& cp, const int32_t* radii, const GsRec* recs, const float* cov3Ds, const uint8_t* clamped,
                             const GsGrad* grads, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dshs_rest,
                             float* dL_dcolors, float* dL_dopac, float* dL_dscales, float* dL_drots, float* dL_dcov3D, float* sh_gate,
                             float* sh_rest_gate) {
  if (P <= 0) return 0;
  const bool posed = g_fused.posed.pose != nullptr;
  const GsPosed pa = posed ? g_fused.posed : GsPosed();
  float* gate = posed ? g_fused.gate : nullptr;
  const int gi[4] = {posed ? g_fused.gate_xyz : -1, posed ? g_fused.gate_rot : -1, posed ? g_fused.gate_scaling : -1,
                     posed ? g_fused.gate_opacity : -1};
#define GS_BWD(POSED, DEG)                                                                                                              \
  hipLaunchKernelGGL((k_preprocess_bwd<POSED, DEG>), dim3((P + 255) / 256), dim3(256), 0, stream, P, M, means3D, shs, shs_rest, scales, \
                     rotations, use_shs, use_cov_precomp, cp, radii, recs, cov3Ds, clamped, grads, dL_dmeans3D, dL_dmeans2D, dL_dshs,  \
                     dL_dshs_rest, dL_dcolors, dL_dopac, dL_dscales, dL_drots, dL_dcov3D, sh_gate, sh_rest_gate, pa, gate, gi[0],     \
                     gi[1], gi[2], gi[3])
  const int deg = use_shs ? D : 0;
  if (posed) { if (deg == 0) GS_BWD(true, 0); else if (deg == 1) GS_BWD(true, 1); else if (deg == 2) GS_BWD(true, 2); else GS_BWD(true, 3); }
  else { if (deg == 0) GS_BWD(false, 0); else if (deg == 1) GS_BWD(false, 1); else if (deg == 2) GS_BWD(false, 2); else GS_BWD(false, 3); }
#undef GS_BWD
  return 0;
}

int gs_launch_mark_visible(hipStream_t stream, int P, const float* means3D, const float* view, uint8_t* present) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, view, present);
  return 0;
}
