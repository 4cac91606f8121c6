// Shared definitions for the gfx950 kernels of libmi355gs.so.
// Scratch-buffer layouts (all offsets 256-byte aligned) are defined here once so that the
// size queries, the forward and the backward agree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mi355gs.h"
#include <type_traits>

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

// Binning (binning.hip): a workgroup counts / scatters GS_BIN_CHUNK consecutive Gaussians; the count kernel leaves each
// workgroup's touched tiles as up to GS_BIN_ENTRIES packed (tile | count << 16) words for the scatter kernel of the same frame.
constexpr int GS_BIN_CHUNK = 512;
constexpr int GS_BIN_ENTRIES = 1024;

struct GeomLayout {
  size_t rec, cov3D, rect, clamped, depth, bin_entries, bin_n, total;
  __host__ explicit GeomLayout(int P) {
    size_t n = P > 0 ? (size_t)P : 1, o = 0;
    rec = o; o += gs_align(n * sizeof(GsRec));
    cov3D = o; o += gs_align(n * 6 * sizeof(float));
    rect = o; o += gs_align(n * sizeof(uint2));
    clamped = o; o += gs_align(n);
    depth = o; o += gs_align(n * sizeof(float));   // the view depth once more, densely: the scatter reads 4 B per Gaussian, not a 48-B record's line
    const size_t chunks = (n + GS_BIN_CHUNK - 1) / GS_BIN_CHUNK;
    bin_entries = o; o += gs_align(chunks * GS_BIN_ENTRIES * sizeof(uint32_t));
    bin_n = o; o += gs_align(chunks * sizeof(uint32_t));
    total = o;
  }
};

constexpr int GS_SORT_SMALL_CAP = 2048;  // keys the tile sort's register network takes in one go (binning.hip)

constexpr int GS_UNIT_LEVELS = 4;
// The rule, shared by the device (k_scan_tiles, from the frame's true instance count) and the host (from the capacity, to
// pick the kernel instantiation and to size buffers): lengthen the units while at least `min_units` of them remain.
__host__ __device__ inline int gs_unit_level_for(long long instances, long long min_units) {
  int level = 0;
  while (level + 1 < GS_UNIT_LEVELS && (instances >> (7 + level)) >= min_units) ++level;   // instances / (2 * 64 << level), instances >= 0
  return level;
}
// Deterministic-backward mode (mi355gs_tune_deterministic, composite.hip): the moments of every (Gaussian, tile) instance go to a
// row of their own and are summed per Gaussian in tile order instead of meeting in float atomics.  It enters the layouts below
// (rows + row indices per instance in `binning`, the per-Gaussian row offsets behind the gate flags in the gradient scratch).
int gs_deterministic();           // api.hip: the knob (or the value a trainer handle pinned for its calls)
void gs_pin_deterministic(int v); // >= 0: this thread sizes and launches with v until it is reset to -1 (trainer.hip)
struct DetScratchLayout {         // behind the 256 bytes of gate flags in the gradient scratch; all uint32
  size_t off, block_sums, total;
  __host__ explicit DetScratchLayout(int P) {
    const size_t n = (size_t)(P > 0 ? P : 1) + 1;
    size_t o = 0;
    off = o; o += gs_align((n + 1) * 4);             // exclusive prefix sums of the rectangles' tile counts: first row of Gaussian g
    block_sums = o; o += gs_align(((n + 4095) / 4096 + 2) * 4);
    total = o;
  }
};
int gs_min_units();               // binning.hip: the mi355gs_tune_min_units knob (or the value a trainer handle pinned for its calls)
void gs_pin_min_units(int v);     // > 0: this thread sizes and launches with v until it is reset to 0 (trainer.hip)
constexpr int GS_MIN_UNITS = 40960;  // lengthen units only while at least this many remain (6-7 rounds of the 6144 resident waves:
                                     // measured at C4, 7.3 M instances: 512-instance units 2.34 ms/view, 256: 2.28, 128: 2.21, 64: 2.24)

struct TilesLayout {
  size_t count, start, cursor, final_T, n_contrib, order, seg_first, part_first, meta, qmax, total;
  int gx, gy, T;
  __host__ TilesLayout(int W, int H) {
    gx = (W + GS_TILE - 1) / GS_TILE; gy = (H + GS_TILE - 1) / GS_TILE; T = gx * gy;
    size_t o = 0, npix = (size_t)W * H;
    count = o; o += gs_align((size_t)T * 4);
    cursor = o; o += gs_align((size_t)T * 4);   // count and cursor are adjacent: one memset clears both
    start = o; o += gs_align(((size_t)T + 1) * 4);
    final_T = o; o += gs_align(npix * 4);
    n_contrib = o; o += gs_align(npix * 4);
    order = o; o += gs_align((size_t)T * 4);
    seg_first = o; o += gs_align(((size_t)T + 1) * 4);  // prefix over tiles of ceil(count / unit length): first unit of a tile
    part_first = o; o += gs_align(((size_t)T + 1) * 4);  // prefix over tiles of "the tile's last unit is shorter than the unit length"
    meta = o; o += gs_align(16);   // [1]: backward units of the frame, [2]: chunks of GS_SEG instances per unit, [3]: short units
    qmax = o; o += gs_align((size_t)T * 4 * 4);   // per tile and 8x8 quadrant: the largest per-pixel contributor count (forward -> backward)
    total = o;
  }
};

// Backward work units: a tile's depth-ordered list is cut into segments of GS_SEG instances.  The forward composite
// leaves each pixel's (transmittance, accumulated colour) at every segment boundary, so the backward can replay the
// segments independently: thousands of equal-sized workgroups that the hardware balances, instead of one workgroup per
// tile that lasts as long as the tile is deep (composite.hip).
constexpr int GS_SEG = 64;
// A unit covers 1, 2, 4 or 8 consecutive 64-instance chunks ("level" 0..3), chosen per frame by k_scan_tiles from the frame's
// true instance count and left in meta[2]: boundaries — and with them the 16 B/pixel boundary records and the 32 B/pixel of
// per-pixel state a unit loads — are only needed every (64 << level) instances.  At C3 (0.75 M instances) that is one chunk; at
// C4 (7.3 M) eight: the boundary state touched per frame falls from 700 MB to under 90 MB.  The host only needs to know whether
// the looping instantiation of the backward can be required (level of the CAPACITY > 0; count <= capacity) and an upper bound
// on the number of units for the buffers.

struct BinningLayout {
  size_t keys, list, unit_tile, bstate, hitmask, det_rows, det_rowidx, total;
  uint32_t max_chunks;  // 64-instance chunks the hit-mask table has room for
  uint32_t max_units;  // table / boundary slots available: an upper bound on the units of any frame with <= R instances
  bool may_loop;       // a frame with this capacity can have units longer than one chunk
  __host__ BinningLayout(int64_t R, int T) {
    size_t n = R > 0 ? (size_t)R : 1, o = 0;
    keys = o; o += gs_align(n * 8);
    list = o; o += gs_align(n * 4);
    // A frame with c <= R instances runs at level L(c) <= L(R) and has at most c / (64 << L(c)) + T units (one partial unit per
    // tile).  While L(c) is below the top level the rule stopped lengthening, so c / (64 << L(c)) < 2 * min_units; at the top
    // level it is at most R / (64 << top).  Small capacities never exceed ceil(R / 64).
    const size_t mu = (size_t)gs_min_units(), top = (size_t)GS_SEG << (GS_UNIT_LEVELS - 1);
    const size_t by_chunks = (n + GS_SEG - 1) / GS_SEG, by_rule = 2 * mu + 1, by_top = (n + top - 1) / top;
    const size_t lim = by_chunks < by_rule ? by_chunks : by_rule;
    may_loop = gs_unit_level_for((long long)n, (long long)mu) > 0;
    max_units = (uint32_t)((lim > by_top ? lim : by_top) + (size_t)(T > 0 ? T : 1));
    unit_tile = o; o += gs_align((size_t)max_units * 16);  // uint4 per unit, in launch order: (tile x | tile y << 16, segment, slot, 0)
    bstate = o; o += gs_align((size_t)max_units * 256 * sizeof(float4));  // per boundary: 256 pixels x (T, C0, C1, C2)
    // Per 64-instance chunk of a tile's list, the four quadrants' hit masks the forward's cull produced (which of the chunk's
    // instances can reach alpha >= 1/255 somewhere in the quadrant): the backward reads them instead of repeating the test
    // (4 x ~60 VALU instructions per unit).  Chunk index = (first unit of the tile + unit) * chunks per unit + chunk, which is
    // below instances / 64 + 8 per tile at every unit length.
    max_chunks = (uint32_t)(by_chunks + 8 * (size_t)(T > 0 ? T : 1) + 8);
    hitmask = o; o += gs_align((size_t)max_chunks * 4 * sizeof(uint64_t));
    det_rows = det_rowidx = o;
    if (gs_deterministic()) {   // one row of twelve floats (a GsGrad) and one row index per instance
      det_rows = o; o += gs_align(n * sizeof(GsGrad));
      det_rowidx = o; o += gs_align(n * 4);
    }
    total = o;
  }
};


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
  float scale_grad_factor;   // backward only: dL/dscale = factor x dL/d(scale_modifier * scale) — 1 (the published operator's
                             // convention, default) or scale_modifier (the true derivative), mi355gs_tune_scale_grad
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

// the same for a double (two 32-bit DPP moves per step): the total lands in lanes 48..63
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double gs_dpp_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));   // lanes without a source read +0.0
}
__device__ __forceinline__ double gs_wave_sum_row3_f64(double v) {
  v += gs_dpp_f64<0xB1>(v);
  v += gs_dpp_f64<0x4E>(v);
  v += gs_dpp_f64<0x141>(v);
  v += gs_dpp_f64<0x140>(v);
  v += gs_dpp_f64<0x142, 0xA>(v);
  v += gs_dpp_f64<0x143, 0xC>(v);
  return v;
}

// Two floats in an aligned VGPR pair: element-wise arithmetic on it compiles to CDNA3/4's full-rate packed-FP32 ops
// (v_pk_add/mul/fma_f32: two lanes' worth of work per issue slot).
typedef float gs_v2f __attribute__((vector_size(8)));
__device__ __forceinline__ gs_v2f gs_fma2(gs_v2f a, gs_v2f b, gs_v2f c) { return a * b + c; }  // contracted: v_pk_fma_f32

// Hides a value's producer from the optimiser (no instruction is emitted): stops it from re-computing a product on
// both sides of a DPP exchange, which costs more VALU ops than the exchange saves.
__device__ __forceinline__ float gs_opaque(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// wave64 maximum, in every lane: four DPP steps inside the rows, then the two row swaps (no LDS crossbar)
__device__ __forceinline__ uint32_t gs_wave_max_u32(uint32_t v) {
  auto dpp = [](uint32_t x, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xF, 0xF, true); };
  v = max(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  v = max(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  v = max(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  v = max(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = max(a[0], a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return max(b[0], b[1]);
}

// wave64 inclusive prefix sum in six DPP adds (row_shr:1/2/4/8 inside the rows of 16, then row_bcast:15 and row_bcast:31 carry
// the row totals on): no LDS crossbar round trip per step (a __shfl_up is a ds_bpermute).  All lanes must be active.
__device__ __forceinline__ uint32_t gs_wave_scan_incl_u32(uint32_t v) {
  auto up = [](uint32_t x, auto ctrl, auto rows) {   // lanes without a source (or in a masked row) receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(rows)::value, 0xF, false);
  };
  using all = std::integral_constant<int, 0xF>;
  v += up(v, std::integral_constant<int, 0x111>{}, all{});   // row_shr:1
  v += up(v, std::integral_constant<int, 0x112>{}, all{});   // row_shr:2
  v += up(v, std::integral_constant<int, 0x114>{}, all{});   // row_shr:4
  v += up(v, std::integral_constant<int, 0x118>{}, all{});   // row_shr:8  -> inclusive scan inside every row
  v += up(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast:15 into rows 1 and 3
  v += up(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});   // row_bcast:31 into rows 2 and 3
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

// Transposed pair steps ACROSS rows with the same swaps: v_permlane16_swap exchanges the odd rows of its first operand
// with the even rows of its second, so (first + second) afterwards holds a summed over each row pair in the even rows and
// b in the odd rows — a pair step in two VALU ops, no selects.  v_permlane32_swap does the same for the wave halves.
__device__ __forceinline__ float gs_pair_reduce_rows16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float gs_pair_reduce_rows32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// physical CU of the calling wave (measurement builds: tools/probe_composite.py)
__device__ __forceinline__ uint32_t gs_physical_cu() {
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [14:13]
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));  // HW_REG_XCC_ID [3:0]
  return ((xcc & 7u) << 6) | (((hw >> 13) & 3u) << 4) | ((hw >> 8) & 15u);
}

extern thread_local bool g_krange_open;
void gs_range_pop();
#define GS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    if (g_krange_open) { g_krange_open = false; gs_range_pop(); }               \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize((hipStream_t)stream); \
    if (e_ != hipSuccess) {                                                     \
      gs_log_error(name, hipGetErrorString(e_));                                \
      return MI355GS_ELAUNCH;                                                   \
    }                                                                           \
  } while (0)

void gs_log_error(const char* where, const char* what);

// roctx range around an entry point's launches (mi355gs_profile_ranges, api.hip): one predictable branch when off
extern bool g_ranges_on;
void gs_range_push(const char* name);
void gs_range_pop();
struct GsRange {
  bool active;
  explicit GsRange(const char* name) : active(g_ranges_on) { if (active) gs_range_push(name); }
  ~GsRange() { if (active) gs_range_pop(); }
  GsRange(const GsRange&) = delete;
  GsRange& operator=(const GsRange&) = delete;
};
#define GS_RANGE() GsRange gs_range_(__func__)
// ... and around ONE launch: GS_KRANGE("name") in front of it, popped by the GS_CHECK_LAUNCH("name") behind it
extern thread_local bool g_krange_open;
#define GS_KRANGE(name) do { if (g_ranges_on) { gs_range_push(name); g_krange_open = true; } } while (0)

// Optional in-library kernel timing (mi355gs_profile_*, api.hip): an event pair around the launches made in its scope, on the
// launch stream, when profiling is on and it is this launch's turn.  kind: include/mi355gs.h mi355gs_profile_read.
struct GsProfScope {
  hipStream_t s; hipEvent_t stop; bool active = false;
  GsProfScope(int kind, hipStream_t stream);
  ~GsProfScope();
};

// Set by the fused train step (trainer.hip) around its calls into the per-operator entry points: the trainer zeroes
// every counter / accumulator of the iteration in ONE prologue launch and asks the operators to skip their own memsets,
// and it collects the "gradient tensor has a non-zero" gate flags for PerPointAdam from the kernels that write the
// gradients (gate[k] > 0  <=>  tensor k of the optimizer's group order has a non-zero gradient) instead of a
// separate pass over all gradients.
struct GsPrologue {  // accumulators of one train step, zeroed by the step's first kernel (k_pose_fwd) instead of memsets
  float4* grad_records = nullptr; size_t n_vec = 0;      // the 48-byte GsGrad records, as float4
  uint32_t* tile_counters = nullptr; int n_counters = 0;  // per-tile count + cursor
  float* g_poses = nullptr; int n_pose = 0;
  float* pose_scratch = nullptr;                           // 32 floats
  float* adam_scratch = nullptr;                           // 8 gate flags
};
// One-call train step: the projection kernels take the RAW parameters (xyz, raw quaternion, log-scale, opacity logit) plus
// the camera pose and apply InstantSplat's camera-frame transform / activations themselves (pose_math.h), and their
// backward goes all the way to the raw-parameter gradients and the 16 pose sums — no k_pose_fwd / k_pose_bwd launches
// and no camera-frame intermediates in HBM.
struct GsPosed {
  const float* pose = nullptr;  // [7] (qw,qx,qy,qz,tx,ty,tz); null = inputs are already in the camera frame
  float* acc = nullptr;         // backward: 16 pose sums (see pose_math.h), zeroed by the caller ...
  float* partial = nullptr;     // ... or, if set, one row of 16 per workgroup of the backward projection kernel (no atomics)
};
struct GsFusedStepHooks {
  bool skip_memsets = false;
  GsPrologue prologue;
  GsPosed posed;
  float* gate = nullptr;   // device float[8] or null
  bool gate_tail = false;  // the flags live right behind the GsGrad records (mi355gs_raster_grad_gate_offset) and are cleared with them
  uint32_t* adam_live = nullptr;  // device uint32[16] persisting across steps (see k_adam_multi) or null
  uint32_t adam_seq = 0;          // launch sequence number (never 0) for adam_live
  // Commit gate of the one-call step (trainer.hip): the frame's true instance count (device word written by the tile scan)
  // and the capacity of the instance buffers.  An Adam launch that finds count > capacity writes NOTHING — the frame dropped
  // instances, its gradients are not the iteration's — so the step can be enqueued whole, before the host has seen the count.
  const uint32_t* commit_count = nullptr;
  unsigned long long commit_capacity = 0;
  uint32_t* commit_poison = nullptr;   // device word, sticky: set by a launch that discarded itself; later gated launches then discard themselves too
  int gate_xyz = -1, gate_rot = -1, gate_scaling = -1, gate_opacity = -1, gate_sh = -1, gate_sh_rest = -1, gate_pose = -1;
};
extern thread_local GsFusedStepHooks g_fused;
