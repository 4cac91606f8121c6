// InstantSplat's camera-frame transform of one Gaussian and its backward, shared by the stand-alone pose kernels
// (pose.hip: the autograd path) and by the projection kernels of the one-call train step (preprocess.hip, POSED
// instantiations), so both paths evaluate the same expressions in the same order.
// Semantics: reference gaussian_renderer/__init__.py:81-103, utils/pose_utils.py:10-104 — the pose quaternion is
// NORMALISED for the rotation of the means but used RAW in the Hamilton product with the (raw) Gaussian quaternions.
#pragma once
#include "common.h"

struct PoseMat {
  float R[9];   // rotation from the normalised quaternion, row-major
  float t[3];
  float q[4];   // raw pose quaternion (w,x,y,z)
  float qn[4];  // normalised
  float inv_norm;
};

__device__ __forceinline__ PoseMat load_pose(const float* __restrict__ pose) {
  PoseMat m;
#pragma unroll
  for (int k = 0; k < 4; ++k) m.q[k] = pose[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) m.t[k] = pose[4 + k];
  const float n = sqrtf(m.q[0] * m.q[0] + m.q[1] * m.q[1] + m.q[2] * m.q[2] + m.q[3] * m.q[3]);
  m.inv_norm = 1.0f / n;
#pragma unroll
  for (int k = 0; k < 4; ++k) m.qn[k] = m.q[k] / n;
  const float r = m.qn[0], x = m.qn[1], y = m.qn[2], z = m.qn[3];
  m.R[0] = 1.f - 2.f * (y * y + z * z); m.R[1] = 2.f * (x * y - r * z); m.R[2] = 2.f * (x * z + r * y);
  m.R[3] = 2.f * (x * y + r * z); m.R[4] = 1.f - 2.f * (x * x + z * z); m.R[5] = 2.f * (y * z - r * x);
  m.R[6] = 2.f * (x * z - r * y); m.R[7] = 2.f * (y * z + r * x); m.R[8] = 1.f - 2.f * (x * x + y * y);
  return m;
}

// ---- forward: raw parameters of Gaussian i -> what the rasterizer consumes
__device__ __forceinline__ float3 pose_mean(const PoseMat& m, float x, float y, float z) {
  return make_float3(m.R[0] * x + m.R[1] * y + m.R[2] * z + m.t[0], m.R[3] * x + m.R[4] * y + m.R[5] * z + m.t[1],
                     m.R[6] * x + m.R[7] * y + m.R[8] * z + m.t[2]);
}
__device__ __forceinline__ float4 pose_rot(const PoseMat& m, float4 g /*(w2,x2,y2,z2)*/) {
  const float w1 = m.q[0], x1 = m.q[1], y1 = m.q[2], z1 = m.q[3];
  float4 o;
  o.x = w1 * g.x - x1 * g.y - y1 * g.z - z1 * g.w;
  o.y = w1 * g.y + x1 * g.x + y1 * g.w - z1 * g.z;
  o.z = w1 * g.z - x1 * g.w + y1 * g.x + z1 * g.y;
  o.w = w1 * g.w + x1 * g.z - y1 * g.y + z1 * g.x;
  return o;
}
__device__ __forceinline__ float pose_scale(float s_log) { return expf(s_log); }
__device__ __forceinline__ float pose_opacity(float logit) { return 1.0f / (1.0f + expf(-logit)); }

// ---- backward of one Gaussian.  a[0..2] += dL/dt, a[3..11] += dL/dR (row-major, g_m (x) xyz), a[12..15] += dL/dq_raw
struct PoseGradOut {
  float d_xyz[3];
  float4 d_rot;
  float d_scaling[3];
  float d_opacity_logit;
};
__device__ __forceinline__ PoseGradOut pose_backward_one(const PoseMat& m, float x, float y, float z, float4 q2, const float* scale,
                                                         float o, float gx, float gy, float gz, float4 gr, const float* g_scale,
                                                         float g_opac, float (&a)[16]) {
  PoseGradOut r;
  r.d_xyz[0] = m.R[0] * gx + m.R[3] * gy + m.R[6] * gz;
  r.d_xyz[1] = m.R[1] * gx + m.R[4] * gy + m.R[7] * gz;
  r.d_xyz[2] = m.R[2] * gx + m.R[5] * gy + m.R[8] * gz;
  a[0] += gx; a[1] += gy; a[2] += gz;
  a[3] += gx * x; a[4] += gx * y; a[5] += gx * z;
  a[6] += gy * x; a[7] += gy * y; a[8] += gy * z;
  a[9] += gz * x; a[10] += gz * y; a[11] += gz * z;
  const float w1 = m.q[0], x1 = m.q[1], y1 = m.q[2], z1 = m.q[3];
  r.d_rot.x = w1 * gr.x + x1 * gr.y + y1 * gr.z + z1 * gr.w;
  r.d_rot.y = -x1 * gr.x + w1 * gr.y + z1 * gr.z - y1 * gr.w;
  r.d_rot.z = -y1 * gr.x - z1 * gr.y + w1 * gr.z + x1 * gr.w;
  r.d_rot.w = -z1 * gr.x + y1 * gr.y - x1 * gr.z + w1 * gr.w;
  a[12] += q2.x * gr.x + q2.y * gr.y + q2.z * gr.z + q2.w * gr.w;
  a[13] += -q2.y * gr.x + q2.x * gr.y - q2.w * gr.z + q2.z * gr.w;
  a[14] += -q2.z * gr.x + q2.w * gr.y + q2.x * gr.z - q2.y * gr.w;
  a[15] += -q2.w * gr.x - q2.z * gr.y + q2.y * gr.z + q2.x * gr.w;
#pragma unroll
  for (int k = 0; k < 3; ++k) r.d_scaling[k] = g_scale[k] * scale[k];
  r.d_opacity_logit = g_opac * o * (1.f - o);
  return r;
}

// 256-thread workgroup: sum the 16 accumulators over the workgroup, then either add them to acc[16] with device float
// atomics, or (partial != null) store them as this workgroup's row of partial[gridDim.x][16] for a later reduction — hundreds
// of workgroups adding to the same 16 addresses serialise at the memory side (that was most of this stage's time).
__device__ __forceinline__ void pose_accumulate(float (&a)[16], float* __restrict__ acc, float* __restrict__ partial,
                                                float (*s_red)[16] /*[4][16]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = gs_wave_sum_row3(a[k]);
    if (lane == 63) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int k = threadIdx.x;
    const float v = (s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k]);
    if (partial) partial[(size_t)blockIdx.x * 16 + k] = v;
    else atomicAdd(&acc[k], v);
  }
}
