// Per-point Adam step for one parameter tensor.
// replaces PerPointAdam.step at reference scene/per_point_adam.py:34-100:
//   - moments are updated only if the WHOLE tensor's gradient norm is > 0 (the reference's `mask`
//     is a 0-dim tensor, per_point_adam.py:62-69), while the parameter update is always applied;
//   - eps is added to sqrt(v) before bias correction (folded into step_size, :72-76);
//   - an optional per-point multiplier scales the step of every element of a point's row (:87-88);
//     the reference computes an adjusted multiplier and discards it (:89), so it stays constant.
// HBM-bound: 16 B read + 12 B written per element (4 B read for a gated-off tensor whose first moment is still zero).
#include <math.h>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_adam(int64_t n, int row, float* __restrict__ param, const float* __restrict__ grad,
                                               float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                               const float* __restrict__ per_point_lr, const float* __restrict__ grad_sumsq,
                                               float step_size, float beta1, float beta2, float eps) {
  const bool update = grad_sumsq ? (*grad_sumsq > 0.f) : true;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float m = exp_avg[i];
    if (!update && m == 0.f) continue;  // gated-off tensor, zero first moment: p - s * (0 / denom) == p bit for bit
    float v = exp_avg_sq[i];
    if (update) {
      const float g = grad[i];
      m = m * beta1 + g * (1.f - beta1);
      v = v * beta2 + g * g * (1.f - beta2);
      exp_avg[i] = m;
      exp_avg_sq[i] = v;
    }
    const float denom = sqrtf(v) + eps;
    const float s = per_point_lr ? step_size * per_point_lr[i / row] : step_size;
    param[i] = param[i] - s * (m / denom);
  }
}

}  // namespace

// ---- multi-tensor variant: all parameter groups of one optimizer step in two launches
// (sum of squared gradients per tensor for the reference's whole-tensor gate, then the update).
namespace {

constexpr int MT_MAX = 8;
constexpr int MT_CHUNK = 2048;  // elements per workgroup (measured on C3: 4096 -> 17.3 us, 2048 -> 15.8, 1024 -> 18.9)

struct MultiAdamArgs {
  int n;
  int first_block[MT_MAX + 1];  // workgroup index ranges per tensor
  long long numel[MT_MAX];
  int row[MT_MAX];
  float* param[MT_MAX];
  const float* grad[MT_MAX];
  float* m[MT_MAX];
  float* v[MT_MAX];
  const float* pplr[MT_MAX];
  float step_size[MT_MAX];
  const float* gate;  // caller-provided gate flags (or null)
  int gidx[MT_MAX];   // >= 0: tensor t's moment update is decided by gate[gidx[t]]; -1: by its own sum of squares, sumsq[t];
                      // -2: a tensor that fits ONE workgroup (numel <= MT_CHUNK) sums its squared gradient itself, in k_adam_multi
  int sq_out[MT_MAX]; // k_adam_sumsq: where tensor t's sum goes (its own index unless the launch covers a subset of a step's tensors)
  int vec4[MT_MAX];  // numel % 4 == 0 and all four arrays 16-byte aligned: 128-bit loads/stores
  // Optional memory of gated-off tensors across launches (fused trainer only).  live[2t] = sequence number of the last
  // launch that scanned tensor t's first moment while gated off (0: none since the last update), live[2t+1] = 1 if any
  // scan since then met a non-zero first moment.  A gated-off tensor that an EARLIER launch scanned completely without
  // meeting one is a no-op element for element (p - s * (0 / denom) == p) and is skipped without reading anything:
  // f_rest before the SH degree is raised costs nothing instead of 4 B per element.
  uint32_t* live;
  uint32_t seq;
  // commit gate (fused trainer only): if *commit_count > commit_capacity the launch writes nothing at all
  // ... and neither does any later gated launch of the same trainer: the first one leaves *commit_poison = 1 (sticky)
  const uint32_t* commit_count;
  unsigned long long commit_capacity;
  uint32_t* commit_poison;
};

__device__ __forceinline__ int mt_find(const MultiAdamArgs& a, int b) {
  int t = 0;
#pragma unroll
  for (int k = 1; k < MT_MAX; ++k) t += (k < a.n && b >= a.first_block[k]) ? 1 : 0;
  return t;
}

__global__ __launch_bounds__(256) void k_adam_sumsq(MultiAdamArgs a, float* __restrict__ sumsq) {
  __shared__ float s_red[4];
  const int t = mt_find(a, blockIdx.x);
  const long long lo = (long long)(blockIdx.x - a.first_block[t]) * MT_CHUNK;
  const long long hi = min(a.numel[t], lo + MT_CHUNK);
  const float* g = a.grad[t];
  float acc = 0.f;
  if (a.vec4[t]) {
    for (long long j = (lo >> 2) + threadIdx.x; j < (hi >> 2); j += 256) {
      const float4 x = reinterpret_cast<const float4*>(g)[j];
      acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const float x = g[i]; acc += x * x; }
  }
  acc = gs_wave_sum_row3(acc);
  if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    if (tot > 0.f) atomicAdd(&sumsq[a.sq_out[t]], tot);
  }
}

__global__ __launch_bounds__(256) void k_adam_multi(MultiAdamArgs a, const float* __restrict__ sumsq, float beta1, float beta2,
                                                     float eps) {
  if (a.commit_count) {   // wave-uniform (scalar loads)
    const bool over = (unsigned long long)*a.commit_count > a.commit_capacity;
    if (over || (a.commit_poison && *a.commit_poison)) {
      if (over && a.commit_poison && blockIdx.x == 0 && threadIdx.x == 0) *a.commit_poison = 1u;
      return;
    }
  }
  const int t = mt_find(a, blockIdx.x);
  const long long lo = (long long)(blockIdx.x - a.first_block[t]) * MT_CHUNK;
  const long long hi = min(a.numel[t], lo + MT_CHUNK);
  bool update;
  if (a.gidx[t] == -2) {
    // the whole tensor belongs to this workgroup: evaluate its gate here (any non-zero gradient element)
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    bool nz = false;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) { const float x = a.grad[t][i]; nz = nz || x * x > 0.f; }   // what sum(x^2) > 0 tests
    if (nz) s_any = 1;   // benign same-value race
    __syncthreads();
    update = s_any != 0;
  } else {
    update = (a.gidx[t] >= 0 ? a.gate[a.gidx[t]] : sumsq[t]) > 0.f;
  }
  if (a.live) {
    const bool first = blockIdx.x == (unsigned)a.first_block[t] && threadIdx.x == 0;
    if (update) {
      if (first) { a.live[2 * t] = 0u; a.live[2 * t + 1] = 0u; }
    } else {
      const uint32_t scanned = a.live[2 * t], nonzero = a.live[2 * t + 1];
      if (scanned != 0u && scanned != a.seq && nonzero == 0u) return;  // written by an earlier launch: complete
      if (first) a.live[2 * t] = a.seq;
    }
  }
  bool met_nonzero = false;
  float* __restrict__ p = a.param[t];
  const float* __restrict__ g = a.grad[t];
  float* __restrict__ mm = a.m[t];
  float* __restrict__ vv = a.v[t];
  const float* __restrict__ pplr = a.pplr[t];
  const int row = a.row[t];
  const float step_size = a.step_size[t];
  auto one = [&](float& pv, float gi, float& m, float& v, long long i) {
    if (update) {
      m = m * beta1 + gi * (1.f - beta1);
      v = v * beta2 + gi * gi * (1.f - beta2);
    }
    const float denom = sqrtf(v) + eps;
    const float s = pplr ? step_size * pplr[i / row] : step_size;
    pv = pv - s * (m / denom);
  };
  if (a.vec4[t]) {
    // 16 bytes per lane and access: a quarter of the memory instructions of the scalar loop for the same bytes
    const long long vlo = lo >> 2, vhi = hi >> 2;  // MT_CHUNK and numel are multiples of 4
    for (long long j = vlo + threadIdx.x; j < vhi; j += 256) {
      float4 m4 = reinterpret_cast<const float4*>(mm)[j];
      // A tensor whose gradient is all zero keeps its moments but still takes the parameter step (the reference's
      // whole-tensor gate).  Where the first moment is zero that step is p - s * (0 / denom) == p bit for bit, so only
      // exp_avg is read: f_rest before the SH degree is raised costs 4 B per element instead of 16.
      if (!update && m4.x == 0.f && m4.y == 0.f && m4.z == 0.f && m4.w == 0.f) continue;
      met_nonzero = true;
      float4 v4 = reinterpret_cast<const float4*>(vv)[j];
      float4 p4 = reinterpret_cast<const float4*>(p)[j];
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (update) g4 = reinterpret_cast<const float4*>(g)[j];
      one(p4.x, g4.x, m4.x, v4.x, 4 * j); one(p4.y, g4.y, m4.y, v4.y, 4 * j + 1);
      one(p4.z, g4.z, m4.z, v4.z, 4 * j + 2); one(p4.w, g4.w, m4.w, v4.w, 4 * j + 3);
      if (update) { reinterpret_cast<float4*>(mm)[j] = m4; reinterpret_cast<float4*>(vv)[j] = v4; }
      reinterpret_cast<float4*>(p)[j] = p4;
    }
    if (a.live && !update && met_nonzero) a.live[2 * t + 1] = 1u;
    return;
  }
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    float m = mm[i];
    if (!update && m == 0.f) continue;  // see the vector path
    met_nonzero = true;
    float v = vv[i], pv = p[i];
    const float gi = update ? g[i] : 0.f;
    one(pv, gi, m, v, i);
    if (update) { mm[i] = m; vv[i] = v; }
    p[i] = pv;
  }
  if (a.live && !update && met_nonzero) a.live[2 * t + 1] = 1u;
}

}  // namespace

extern "C" int mi355gs_adam_multi_step(void* stream_, int ntensors, const int64_t* numel, const int32_t* row, float* const* params,
                                       const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                       const float* const* per_point_lr, const float* lr, float beta1, float beta2, float eps,
                                       const int32_t* step, float* scratch, const float* gate, const int32_t* gate_index,
                                       uint32_t* live, uint32_t seq) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (ntensors < 0 || ntensors > MT_MAX || (gate && !gate_index)) return MI355GS_EINVAL;
  if (ntensors == 0) return MI355GS_OK;
  if (!numel || !row || !params || !grads || !exp_avg || !exp_avg_sq || !per_point_lr || !lr || !step) return MI355GS_EINVAL;
  MultiAdamArgs a;
  a.n = ntensors; a.gate = gate;
  int blocks = 0;
  bool any_ungated = false;   // some tensor's gate has to be computed from its gradient
  bool need_sumsq = false;    // ... by the separate pass over the gradients
  for (int t = 0; t < MT_MAX; ++t) {
    a.first_block[t] = blocks;
    if (t < ntensors) {
      if (numel[t] < 0 || row[t] <= 0 || step[t] <= 0 || (numel[t] > 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t])))
        return MI355GS_EINVAL;
      if (gate && gate_index[t] >= MT_MAX) return MI355GS_EINVAL;
      // every negative index means "no flag for this tensor: derive its gate from the gradient" (-1 in the header); -2 is this
      // function's own marker for a tensor gated by the workgroup that owns it and must not be reachable from outside
      a.numel[t] = numel[t]; a.row[t] = row[t]; a.gidx[t] = (gate && gate_index[t] >= 0) ? gate_index[t] : -1; a.sq_out[t] = t;
      if (a.gidx[t] < 0) any_ungated = true;
      a.param[t] = params[t]; a.grad[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.pplr[t] = per_point_lr[t];
      const double bc1 = 1.0 - pow((double)beta1, (double)step[t]), bc2 = 1.0 - pow((double)beta2, (double)step[t]);
      a.step_size[t] = (float)((double)lr[t] * (sqrt(bc2) / bc1));
      a.vec4[t] = (numel[t] % 4 == 0 && (((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avg[t] | (uintptr_t)exp_avg_sq[t]) & 15) == 0) ? 1 : 0;
      blocks += (int)((numel[t] + MT_CHUNK - 1) / MT_CHUNK);
    } else {
      a.numel[t] = 0; a.row[t] = 1; a.param[t] = nullptr; a.grad[t] = nullptr; a.m[t] = nullptr; a.v[t] = nullptr; a.pplr[t] = nullptr;
      a.step_size[t] = 0.f; a.vec4[t] = 0; a.gidx[t] = -1; a.sq_out[t] = t;
    }
  }
  a.first_block[MT_MAX] = blocks;
  a.live = nullptr; a.seq = 0; a.commit_count = nullptr; a.commit_capacity = 0; a.commit_poison = nullptr;
  if (blocks == 0) return MI355GS_OK;
  if (!gate && g_fused.gate == scratch) {
    a.live = g_fused.adam_live; a.seq = g_fused.adam_seq;
    a.commit_count = g_fused.commit_count; a.commit_capacity = g_fused.commit_capacity; a.commit_poison = g_fused.commit_poison;
  } else if (live && seq != 0u) {
    a.live = live; a.seq = seq;   // the caller's own memory of gated-off tensors (include/mi355gs.h)
  }
  if (gate && any_ungated) {
    // small leftovers (the pose table: 7 V floats) are gated inside the update kernel, by the one workgroup that owns them
    for (int t = 0; t < ntensors; ++t) {
      if (a.gidx[t] >= 0) continue;
      if (a.numel[t] <= MT_CHUNK) a.gidx[t] = -2; else need_sumsq = true;
    }
  } else if (any_ungated) {
    need_sumsq = true;
  }
  if (need_sumsq && !scratch) return MI355GS_EINVAL;
  if (gate) {
    // the caller vouches that gate[gate_index[t]] > 0 <=> grads[t] has a non-zero element (the flags
    // mi355gs_posed_backward leaves behind the gradient records): no pass over those gradients.  Tensors without a flag
    // (gate_index[t] < 0: e.g. the pose table, whose gradient autograd scatters from the one row the backward wrote) are
    // summed by a launch that covers only them.
    if (need_sumsq) {
      MultiAdamArgs s = a;
      int k = 0, sblocks = 0;
      for (int t = 0; t < ntensors; ++t) {
        if (a.gidx[t] != -1) continue;
        s.first_block[k] = sblocks; s.numel[k] = a.numel[t]; s.grad[k] = a.grad[t]; s.vec4[k] = a.vec4[t]; s.sq_out[k] = t;
        sblocks += (int)((a.numel[t] + MT_CHUNK - 1) / MT_CHUNK);
        ++k;
      }
      for (int j = k; j <= MT_MAX; ++j) s.first_block[j] = sblocks;
      s.n = k;
      if (hipMemsetAsync(scratch, 0, MT_MAX * sizeof(float), stream) != hipSuccess) return MI355GS_ELAUNCH;
      if (sblocks > 0) {
        GS_KRANGE("adam_sumsq");
        hipLaunchKernelGGL(k_adam_sumsq, dim3(sblocks), dim3(256), 0, stream, s, scratch);
        GS_CHECK_LAUNCH("adam_sumsq");
      }
    }
  } else if (g_fused.gate == scratch) {
    // fused train step: the gate flags were written by the kernels that produced the gradients
    a.gate = scratch;
    for (int t = 0; t < ntensors; ++t) a.gidx[t] = t;
  } else {
    if (hipMemsetAsync(scratch, 0, MT_MAX * sizeof(float), stream) != hipSuccess) return MI355GS_ELAUNCH;
    GS_KRANGE("adam_sumsq");
    hipLaunchKernelGGL(k_adam_sumsq, dim3(blocks), dim3(256), 0, stream, a, scratch);
    GS_CHECK_LAUNCH("adam_sumsq");
  }
  GS_KRANGE("adam_multi");
  hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, stream, a, (const float*)scratch, beta1, beta2, eps);
  GS_CHECK_LAUNCH("adam_multi");
  return MI355GS_OK;
}

extern "C" int mi355gs_adam_step(void* stream_, int64_t n, int row, float* param, const float* grad, float* exp_avg,
                                 float* exp_avg_sq, const float* per_point_lr, const float* grad_sumsq, float lr, float beta1,
                                 float beta2, float eps, int step) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (n < 0 || row <= 0 || step <= 0 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return MI355GS_EINVAL;
  if (n == 0) return MI355GS_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * (sqrt(bc2) / bc1));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048 * 4) blocks = 2048 * 4;
  GS_KRANGE("adam");
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, stream, n, row, param, grad, exp_avg, exp_avg_sq, per_point_lr,
                     grad_sumsq, step_size, beta1, beta2, eps);
  GS_CHECK_LAUNCH("adam");
  return MI355GS_OK;
}
