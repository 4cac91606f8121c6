// Shared definitions for the gfx950 kernels of libmi355gs.so.
// Scratch-buffer layouts (all offsets 256-byte aligned) are defined here once so that the
// size queries, the forward and the backward agree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mi355gs.h"

#define GS_LOG2E 1.4426950408889634f
#define GS_TILE 16
#define GS_TILE_PIX 256
#define GS_WAVE 64

// One 48-byte record per Gaussian, written by preprocess and gathered by the composite kernels
// with three 16-byte loads from one contiguous address (1 cache line most of the time).
struct alignas(16) GsRec {
  float4 q0;  // x, y (pixel centre), hx, hy (half extents of the alpha >= 1/255 ellipse's AABB; <0: never visible)
  float4 q1;  // conic pre-scaled for v_exp_f32 and ordered for packed-FP32 math: (-a/2, -c/2, -b) * log2(e), then
              // opacity — so that G = exp2(q1.x*dx*dx + q1.y*dy*dy + q1.z*dx*dy) with no further multiplies and
              // (q1.x, q1.y) * (dx, dy) is one v_pk_mul_f32
  float4 q2;  // r, g, b, depth
};

// Per-Gaussian accumulator filled by composite backward (float atomics): moments of w = G * dL/dG over all
// pixels the Gaussian touched, plus the colour gradient.  k_preprocess_bwd turns the moments into
// dL/dmean2D, dL/dconic and dL/dopacity with the Gaussian's own conic and opacity.
struct alignas(16) GsGrad {
  float4 g0;  // sum w dx, sum w dy, sum w dx^2, sum w dx dy
  float4 g1;  // sum w dy^2, sum w, dL/dr, dL/dg
  float4 g2;  // dL/db, unused x3
};

static inline size_t gs_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct GeomLayout {
  size_t rec, cov3D, rect, clamped, total;
  __host__ explicit GeomLayout(int P) {
    size_t n = P > 0 ? (size_t)P : 1, o = 0;
    rec = o; o += gs_align(n * sizeof(GsRec));
    cov3D = o; o += gs_align(n * 6 * sizeof(float));
    rect = o; o += gs_align(n * sizeof(uint2));
    clamped = o; o += gs_align(n);
    total = o;
  }
};

struct TilesLayout {
  size_t count, start, cursor, final_T, n_contrib, total;
  int gx, gy, T;
  __host__ TilesLayout(int W, int H) {
    gx = (W + GS_TILE - 1) / GS_TILE; gy = (H + GS_TILE - 1) / GS_TILE; T = gx * gy;
    size_t o = 0, npix = (size_t)W * H;
    count = o; o += gs_align((size_t)T * 4);
    cursor = o; o += gs_align((size_t)T * 4);   // count and cursor are adjacent: one memset clears both
    start = o; o += gs_align(((size_t)T + 1) * 4);
    final_T = o; o += gs_align(npix * 4);
    n_contrib = o; o += gs_align(npix * 4);
    total = o;
  }
};

struct BinningLayout {
  size_t keys, list, total;
  __host__ explicit BinningLayout(int64_t R) {
    size_t n = R > 0 ? (size_t)R : 1, o = 0;
    keys = o; o += gs_align(n * 8);
    list = o; o += gs_align(n * 4);
    total = o;
  }
};

// XCD-aware block -> tile map: hardware places block b on XCD b % 8 (observed, speed only); give
// each XCD one contiguous band of tile rows so neighbouring tiles (which share Gaussians) share an L2.
__device__ __forceinline__ int gs_tile_of_block(int b, int T) {
  const int per = (T + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}
static inline int gs_grid_for_tiles(int T) { return ((T + 7) >> 3) << 3; }

// transposed (column-major flat) 4x4 helpers, the storage the reference hands over
__device__ __forceinline__ float3 gs_tp43(const float* m, float3 p) {
  return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 gs_tp44(const float* m, float3 p) {
  return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

struct CamParams {  // passed by value (kernarg): wave-uniform, lives in SGPRs
  const float* view;    // device pointers: uniform address -> scalar loads
  const float* proj;
  const float* campos;
  float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
  int W, H, gx, gy;
};

// wave64 sum via butterfly shuffles; every lane gets the total
__device__ __forceinline__ float gs_wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// DPP (data-parallel primitive) lane exchange: one VALU op, no LDS traffic.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float gs_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// wave64 sum in 6 DPP adds; the total lands in lanes 48..63 (the last row of 16). All lanes must be active.
__device__ __forceinline__ float gs_wave_sum_row3(float v) {
  v += gs_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  v += gs_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
  v += gs_dpp<0x141>(v);       // row_half_mirror
  v += gs_dpp<0x140>(v);       // row_mirror        -> every lane holds its row's sum
  v += gs_dpp<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v += gs_dpp<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> row 3 holds the wave total
  return v;
}

// Two floats in an aligned VGPR pair: element-wise arithmetic on it compiles to CDNA3/4's full-rate packed-FP32 ops
// (v_pk_add/mul/fma_f32: two lanes' worth of work per issue slot).
typedef float gs_v2f __attribute__((vector_size(8)));

// Hides a value's producer from the optimiser (no instruction is emitted): stops it from re-computing a product on
// both sides of a DPP exchange, which costs more VALU ops than the exchange saves.
__device__ __forceinline__ float gs_opaque(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Transposed pair step of a multi-value wave reduction: lanes whose `bit` is clear keep `a`, the others keep
// `b`; each lane adds its exchange partner's copy of the value it keeps.  Two values in, one out, and the
// surviving value differs per lane — so after log2 steps nine values are reduced with ~1/3 of the DPP traffic
// of nine independent butterflies.  CTRL must pair lanes that differ in exactly that bit.
template <int CTRL>
__device__ __forceinline__ float gs_pair_reduce(bool bit, float a, float b) {
  const float keep = bit ? b : a, send = bit ? a : b;
  return keep + gs_dpp<CTRL>(send);
}

// Sum over the four 16-lane rows, lane-wise (every lane ends with the sum of the lanes l, l^16, l^32, l^48), with
// gfx950's VALU row swaps — no LDS round trip (ds_bpermute) on the critical path.
__device__ __forceinline__ float gs_sum_rows(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned w = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

#define GS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize((hipStream_t)stream); \
    if (e_ != hipSuccess) {                                                     \
      gs_log_error(name, hipGetErrorString(e_));                                \
      return MI355GS_ELAUNCH;                                                   \
    }                                                                           \
  } while (0)

void gs_log_error(const char* where, const char* what);

// Set by the fused train step (trainer.hip) around its calls into the per-operator entry points: the trainer zeroes
// every counter / accumulator of the iteration in ONE prologue launch and asks the operators to skip their own memsets,
// and it collects the "gradient tensor has a non-zero" gate flags for PerPointAdam from the kernels that write the
// gradients (gate[k] > 0  <=>  tensor k of the optimizer's group order has a non-zero gradient) instead of a
// separate pass over all gradients.
struct GsFusedStepHooks {
  bool skip_memsets = false;
  float* gate = nullptr;   // device float[8] or null
  int gate_xyz = -1, gate_rot = -1, gate_scaling = -1, gate_opacity = -1, gate_sh = -1, gate_sh_rest = -1, gate_pose = -1;
};
extern thread_local GsFusedStepHooks g_fused;
