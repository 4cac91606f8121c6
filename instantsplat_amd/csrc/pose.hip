// InstantSplat's camera-frame transform, fused with the Gaussian activations.
//
// replaces, per render call, the ~60 eager PyTorch kernels of reference
// gaussian_renderer/__init__.py:81-103 (get_camera_from_tensor -> [R|t], xyz_cam = R xyz + t,
// rot_cam = pose_quat (x) rotation, opacity = sigmoid, scales = exp; helpers at
// utils/pose_utils.py:10-104, scene/gaussian_model.py:101-124) and, in backward, their autograd
// graph plus the P-long reduction to the 7 pose gradients — one launch each way.
// Semantics kept: the pose quaternion is NORMALISED for the rotation of the means but used RAW in
// the Hamilton product with the (raw) Gaussian quaternions.
// HBM-bound: 44 B read + 44 B written per Gaussian forward; 88 B read + 44 B written backward.
#include "pose_math.h"

namespace {

__global__ __launch_bounds__(256) void k_pose_fwd(int P, const float* __restrict__ xyz, const float* __restrict__ rot,
                                                   const float* __restrict__ scaling, const float* __restrict__ opacity_logit,
                                                   const float* __restrict__ pose, float* __restrict__ means_cam,
                                                   float* __restrict__ rot_cam, float* __restrict__ scales, float* __restrict__ opac,
                                                   GsPrologue pro) {
  const PoseMat m = load_pose(pose);
  const int stride = gridDim.x * blockDim.x;
  if (pro.grad_records) {  // first kernel of a fused train step: clear the step's accumulators on the way
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = gid; i < pro.n_vec; i += (size_t)stride) pro.grad_records[i] = z;
    for (size_t i = gid; i < (size_t)pro.n_counters; i += (size_t)stride) pro.tile_counters[i] = 0u;
    if (gid < (size_t)pro.n_pose) pro.g_poses[gid] = 0.f;
    if (gid < 32) pro.pose_scratch[gid] = 0.f;
    if (gid < 8) pro.adam_scratch[gid] = 0.f;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
    const float3 mc = pose_mean(m, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
    means_cam[3 * (size_t)i] = mc.x; means_cam[3 * (size_t)i + 1] = mc.y; means_cam[3 * (size_t)i + 2] = mc.z;
    *reinterpret_cast<float4*>(rot_cam + 4 * (size_t)i) = pose_rot(m, *reinterpret_cast<const float4*>(rot + 4 * (size_t)i));
#pragma unroll
    for (int k = 0; k < 3; ++k) scales[3 * (size_t)i + k] = pose_scale(scaling[3 * (size_t)i + k]);
    opac[i] = pose_opacity(opacity_logit[i]);
  }
}

// acc -> dL/dpose (run by one thread once every workgroup's partial sums are in)
__device__ __forceinline__ void pose_finish(const float* __restrict__ pose, const float* acc, float* __restrict__ d_pose,
                                            float* __restrict__ pose_gate) {
  const PoseMat m = load_pose(pose);
  const float* Rp = acc + 3;  // dL/dR, row-major: Rp[3*i+j]
  const float r = m.qn[0], x = m.qn[1], y = m.qn[2], z = m.qn[3];
  float gq[4];
  gq[0] = 2.f * (z * (Rp[3] - Rp[1]) + y * (Rp[2] - Rp[6]) + x * (Rp[7] - Rp[5]));
  gq[1] = 2.f * (y * (Rp[1] + Rp[3]) + z * (Rp[2] + Rp[6]) + r * (Rp[7] - Rp[5])) - 4.f * x * (Rp[4] + Rp[8]);
  gq[2] = 2.f * (x * (Rp[1] + Rp[3]) + r * (Rp[2] - Rp[6]) + z * (Rp[5] + Rp[7])) - 4.f * y * (Rp[0] + Rp[8]);
  gq[3] = 2.f * (r * (Rp[3] - Rp[1]) + x * (Rp[2] + Rp[6]) + y * (Rp[5] + Rp[7])) - 4.f * z * (Rp[0] + Rp[4]);
  // through q_hat = q / |q|
  const float dot = m.qn[0] * gq[0] + m.qn[1] * gq[1] + m.qn[2] * gq[2] + m.qn[3] * gq[3];
  float d[7];
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = (gq[k] - m.qn[k] * dot) * m.inv_norm + acc[12 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) d[4 + k] = acc[k];
  bool nz = false;
#pragma unroll
  for (int k = 0; k < 7; ++k) { d_pose[k] = d[k]; nz = nz || d[k] != 0.f; }
  if (pose_gate && nz) *pose_gate = 1.0f;
}

__global__ void k_pose_finish(const float* __restrict__ pose, const float* __restrict__ acc, float* __restrict__ d_pose,
                              float* __restrict__ pose_gate) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  pose_finish(pose, acc, d_pose, pose_gate);
}

// acc[0..2] = dL/dt, acc[3..11] = dL/dR (row-major, sum of g_m (x) xyz), acc[12..15] = dL/dq_raw via the Hamilton product
__global__ __launch_bounds__(256) void k_pose_bwd(int P, const float* __restrict__ xyz, const float* __restrict__ rot,
                                                   const float* __restrict__ scales, const float* __restrict__ opac,
                                                   const float* __restrict__ pose, const float* __restrict__ g_means,
                                                   const float* __restrict__ g_rot, const float* __restrict__ g_scales,
                                                   const float* __restrict__ g_opac, float* __restrict__ d_xyz,
                                                   float* __restrict__ d_rot, float* __restrict__ d_scaling,
                                                   float* __restrict__ d_opacity_logit, float* __restrict__ acc,
                                                   float* __restrict__ gate, int gi_xyz, int gi_rot, int gi_scaling, int gi_opacity) {
  __shared__ float s_red[4][16];
  const PoseMat m = load_pose(pose);
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
  bool nz_xyz = false, nz_rot = false, nz_sc = false, nz_op = false;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
    const PoseGradOut r = pose_backward_one(
        m, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], *reinterpret_cast<const float4*>(rot + 4 * (size_t)i),
        scales + 3 * (size_t)i, opac[i], g_means[3 * (size_t)i], g_means[3 * (size_t)i + 1], g_means[3 * (size_t)i + 2],
        *reinterpret_cast<const float4*>(g_rot + 4 * (size_t)i), g_scales + 3 * (size_t)i, g_opac[i], a);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d_xyz[3 * (size_t)i + k] = r.d_xyz[k];
      d_scaling[3 * (size_t)i + k] = r.d_scaling[k];
      nz_xyz = nz_xyz || r.d_xyz[k] != 0.f;
      nz_sc = nz_sc || r.d_scaling[k] != 0.f;
    }
    *reinterpret_cast<float4*>(d_rot + 4 * (size_t)i) = r.d_rot;
    nz_rot = nz_rot || r.d_rot.x != 0.f || r.d_rot.y != 0.f || r.d_rot.z != 0.f || r.d_rot.w != 0.f;
    d_opacity_logit[i] = r.d_opacity_logit;
    nz_op = nz_op || r.d_opacity_logit != 0.f;
  }
  pose_accumulate(a, acc, nullptr, s_red);
  if (gate) {  // PerPointAdam's whole-tensor gate: any non-zero gradient element (benign same-value store race)
    if (gi_xyz >= 0 && nz_xyz) gate[gi_xyz] = 1.0f;
    if (gi_rot >= 0 && nz_rot) gate[gi_rot] = 1.0f;
    if (gi_scaling >= 0 && nz_sc) gate[gi_scaling] = 1.0f;
    if (gi_opacity >= 0 && nz_op) gate[gi_opacity] = 1.0f;
  }
}

__global__ __launch_bounds__(1024) void k_pose_finish_partials(const float* __restrict__ pose, const float* __restrict__ partial, int nrows,
                                                               float* __restrict__ d_pose, float* __restrict__ pose_gate,
                                                               const float* __restrict__ loss_partial, int loss_nblocks,
                                                               double loss_inv_n, float lambda_dssim, float* __restrict__ loss,
                                                               int table_rows, int table_row) {
  // d_pose may be a whole [table_rows, 7] pose-table gradient (what autograd's select-backward builds with a fill and a copy
  // from the 7 values): row table_row gets the gradient, every other row is zeroed here
  if (table_rows > 0) {
    for (int i = threadIdx.x; i < table_rows * 7; i += 1024)
      if (i / 7 != table_row) d_pose[i] = 0.f;
    d_pose += 7 * table_row;
  }
  __shared__ float s_sum[16][17];
  __shared__ float s_tot[16];
  // the rows of pose sums first: their loads are in flight while the loss partials are reduced (this kernel is pure latency)
  const int k = threadIdx.x & 15, g = threadIdx.x >> 4;  // 64 row groups x 16 sums; a wave reads 4 consecutive rows (256 B)
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  {
    int r = g;
    for (; r + 192 < nrows; r += 256) {
      v0 += partial[(size_t)r * 16 + k]; v1 += partial[(size_t)(r + 64) * 16 + k];
      v2 += partial[(size_t)(r + 128) * 16 + k]; v3 += partial[(size_t)(r + 192) * 16 + k];
    }
    for (; r < nrows; r += 64) v0 += partial[(size_t)r * 16 + k];
  }
  // One-call train step: this single-workgroup, latency-bound kernel also sums the loss kernel's per-workgroup partials into the
  // loss VALUE (reference train.py:176; only ever read by the host) — in a fixed order, double accumulation — instead of a
  // finishing launch of its own.
  if (loss) {
    __shared__ double s_la[16], s_lb[16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < loss_nblocks; i += 1024) { a += (double)loss_partial[2 * i]; b += (double)loss_partial[2 * i + 1]; }
    a = gs_wave_sum_row3_f64(a); b = gs_wave_sum_row3_f64(b);   // DPP: a 64-bit shuffle butterfly is twelve LDS-crossbar round trips per value
    if ((threadIdx.x & 63) == 63) { s_la[threadIdx.x >> 6] = a; s_lb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0.0, tb = 0.0;
      for (int w = 0; w < 16; ++w) { ta += s_la[w]; tb += s_lb[w]; }
      *loss = (1.0f - lambda_dssim) * (float)(tb * loss_inv_n) + lambda_dssim * (1.0f - (float)(ta * loss_inv_n));
    }
  }
  // a wave holds four row groups (its four rows of 16 lanes): add them with the row swaps, then 16 waves x 16 sums through LDS
  const float wsum = gs_sum_rows((v0 + v1) + (v2 + v3));
  if ((threadIdx.x & 63) < 16) s_sum[threadIdx.x >> 6][k] = wsum;
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_sum[q][threadIdx.x];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) total[c] = s_tot[c];
    pose_finish(pose, total, d_pose, pose_gate);
  }
}

}  // namespace

// dL/dpose from per-workgroup rows of the 16 pose sums (internal entry for the one-call train step, whose backward
// projection kernel stores one row per workgroup instead of issuing atomics): deterministic tree sum, then pose_finish
int gs_launch_pose_finish_partials(hipStream_t stream, const float* pose, const float* partial, int nrows, float* d_pose,
                                   float* pose_gate, const float* loss_partial, int loss_nblocks, double loss_inv_n,
                                   float lambda_dssim, float* loss, int table_rows, int table_row) {
  hipLaunchKernelGGL(k_pose_finish_partials, dim3(1), dim3(1024), 0, stream, pose, partial, nrows, d_pose, pose_gate, loss_partial,
                     loss_nblocks, loss_inv_n, lambda_dssim, loss, table_rows, table_row);
  return 0;
}

extern "C" {

int mi355gs_pose_forward(void* stream_, int P, const float* xyz, const float* rot, const float* scaling,
                         const float* opacity_logit, const float* pose, float* means_cam, float* rot_cam, float* scales,
                         float* opac) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (P < 0 || !pose) return MI355GS_EINVAL;
  if (P == 0) return MI355GS_OK;
  if (!xyz || !rot || !scaling || !opacity_logit || !means_cam || !rot_cam || !scales || !opac) return MI355GS_EINVAL;
  const int blocks = min((P + 255) / 256, 4096);
  GS_KRANGE("pose_fwd");
  hipLaunchKernelGGL(k_pose_fwd, dim3(blocks), dim3(256), 0, stream, P, xyz, rot, scaling, opacity_logit, pose, means_cam, rot_cam,
                     scales, opac, g_fused.prologue);
  GS_CHECK_LAUNCH("pose_fwd");
  return MI355GS_OK;
}

int mi355gs_pose_backward(void* stream_, int P, const float* xyz, const float* rot, const float* scales, const float* opac,
                          const float* pose, const float* g_means, const float* g_rot, const float* g_scales,
                          const float* g_opac, float* d_xyz, float* d_rot, float* d_scaling, float* d_opacity_logit,
                          float* d_pose, float* scratch16) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (P < 0 || !pose || !d_pose || !scratch16) return MI355GS_EINVAL;
  if (!g_fused.skip_memsets && hipMemsetAsync(scratch16, 0, 32 * sizeof(float), stream) != hipSuccess) return MI355GS_ELAUNCH;
  float* pose_gate = (g_fused.gate && g_fused.gate_pose >= 0) ? g_fused.gate + g_fused.gate_pose : nullptr;
  if (P > 0) {
    if (!xyz || !rot || !scales || !opac || !g_means || !g_rot || !g_scales || !g_opac || !d_xyz || !d_rot || !d_scaling ||
        !d_opacity_logit)
      return MI355GS_EINVAL;
    const int blocks = min((P + 255) / 256, 1024);
    GS_KRANGE("pose_bwd");
    hipLaunchKernelGGL(k_pose_bwd, dim3(blocks), dim3(256), 0, stream, P, xyz, rot, scales, opac, pose, g_means, g_rot, g_scales,
                       g_opac, d_xyz, d_rot, d_scaling, d_opacity_logit, scratch16, g_fused.gate, g_fused.gate_xyz, g_fused.gate_rot,
                       g_fused.gate_scaling, g_fused.gate_opacity);
    GS_CHECK_LAUNCH("pose_bwd");
  }
  // (folding this into k_pose_bwd's last workgroup was measured: the ticket round trips cost more than the launch)
  GS_KRANGE("pose_finish");
  hipLaunchKernelGGL(k_pose_finish, dim3(1), dim3(64), 0, stream, pose, (const float*)scratch16, d_pose, pose_gate);
  GS_CHECK_LAUNCH("pose_finish");
  return MI355GS_OK;
}

}  // extern "C"
