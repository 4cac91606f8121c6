// extern "C" entry points of libmi355gs.so (declared in include/mi355gs.h).
// Host-side only: argument checks, scratch-layout arithmetic, kernel enqueue order.
#include <stdio.h>
#include "common.h"
#include <dlfcn.h>
#include <cstdlib>

// launchers implemented next to their kernels
int gs_launch_preprocess_fwd(hipStream_t, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*,
                             const float*, const float*, const CamParams&, int32_t*, GsRec*, float*, uint2*, uint8_t*, float*, uint8_t*,
                             const GsPrologue&);
int gs_launch_count_tiles(hipStream_t, int, int, int, const uint2*, uint32_t*, uint32_t*, uint32_t*);
int gs_launch_preprocess_bwd(hipStream_t, int, int, int, const float*, const float*, const float*, const float*, const float*, int, int,
                             const CamParams&, const int32_t*, const GsRec*, const float*, const uint8_t*, const GsGrad*, float*, float*,
                             float*, float*, float*, float*, float*, float*, float*, float*, float*);
int gs_launch_mark_visible(hipStream_t, int, const float*, const float*, uint8_t*);
int gs_launch_scan_tiles(hipStream_t, int, const uint32_t*, uint32_t*, int32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
int gs_launch_binning(hipStream_t, int, int, int, const float*, const uint2*, const uint32_t*, uint32_t*, uint64_t*, uint32_t*, uint32_t,
                      const uint32_t*, const uint32_t*, const uint32_t*);
int gs_launch_composite_fwd(hipStream_t, int, int, int, int, uint32_t, const uint32_t*, const uint32_t*, const GsRec*, const float*,
                            float*, float*, uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint4*, float4*, uint32_t,
                            const uint32_t*, unsigned long long*, uint32_t, uint32_t*, unsigned long long*, bool);
int gs_launch_composite_bwd(hipStream_t, int, int, int, uint32_t, const uint32_t*, const uint32_t*, const GsRec*, const float*,
                            const float*, const uint32_t*, const float*, GsGrad*, const float*, const uint4*,
                            const float4*, const uint32_t*, uint32_t, bool, const unsigned long long*, uint32_t, const uint32_t*,
                            unsigned long long*, const uint32_t*, float*);
int gs_launch_det_prepare(hipStream_t, int, int, int, uint32_t, const uint32_t*, const uint32_t*, const uint2*, char*, uint32_t*, float*);
int gs_launch_det_gather(hipStream_t, int, uint32_t, const char*, const float*, GsGrad*);
int gs_launch_frame_stats(hipStream_t, int, int, int, int, const uint32_t*, const uint32_t*, int64_t*);

// ---- optional per-kernel timing (HIP events on the launch stream)
namespace {
constexpr int PROF_KINDS = 6, PROF_MAX = 8192;   // include/mi355gs.h, mi355gs_profile_read: composite forward / backward / render-only
                                                 // forward, fused L1+SSIM loss pass, per-tile sort, per-tile count
struct ProfState {
  bool on = false;
  unsigned long long* work_counters = nullptr;   // device uint64[16] or null (mi355gs_profile_work_counters)
  hipEvent_t ev[PROF_KINDS][PROF_MAX][2];
  int created[PROF_KINDS] = {};
  int used[PROF_KINDS] = {};
  int period = 1;                   // events go around every period-th launch of a kind (mi355gs_profile_set_period)
  int seen[PROF_KINDS] = {};       // launches of the kind since profile_begin
} g_prof;
}  // namespace

GsProfScope::GsProfScope(int kind, hipStream_t stream) : s(stream) {
  if (!g_prof.on || kind < 0 || kind >= PROF_KINDS || g_prof.used[kind] >= PROF_MAX) return;
  if (g_prof.seen[kind]++ % g_prof.period != 0) return;
  const int i = g_prof.used[kind];
  if (i >= g_prof.created[kind]) {
    if (hipEventCreate(&g_prof.ev[kind][i][0]) != hipSuccess || hipEventCreate(&g_prof.ev[kind][i][1]) != hipSuccess) return;
    g_prof.created[kind] = i + 1;
  }
  (void)hipEventRecord(g_prof.ev[kind][i][0], s);
  stop = g_prof.ev[kind][i][1];
  g_prof.used[kind] = i + 1;
  active = true;
}
GsProfScope::~GsProfScope() { if (active) (void)hipEventRecord(stop, s); }
typedef GsProfScope ProfScope;

thread_local GsFusedStepHooks g_fused;

// ---- roctx ranges (mi355gs_profile_ranges): the marker library is opened at run time, on request
bool g_ranges_on = false;
thread_local bool g_krange_open = false;
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
  unsigned long long pushed = 0;
} g_roctx;
bool roctx_open() {
  if (!g_roctx.tried) {
    g_roctx.tried = true;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (g_roctx.push && g_roctx.pop) break;
      g_roctx.push = nullptr; g_roctx.pop = nullptr;
    }
  }
  return g_roctx.push != nullptr;
}
struct RoctxEnv {   // MI355GS_ROCTX=1: ranges from the first call on (a whole run under rocprofv3 --marker-trace)
  RoctxEnv() { const char* e = getenv("MI355GS_ROCTX"); if (e && e[0] == '1' && roctx_open()) g_ranges_on = true; }
} g_roctx_env;
}  // namespace
void gs_range_push(const char* name) { ++g_roctx.pushed; (void)g_roctx.push(name); }
void gs_range_pop() { (void)g_roctx.pop(); }

void gs_log_error(const char* where, const char* what) { fprintf(stderr, "[mi355gs] %s failed: %s\n", where, what); }

static int g_scale_grad_exact = 0;   // mi355gs_tune_scale_grad
static int g_deterministic = 0;      // mi355gs_tune_deterministic
thread_local int g_deterministic_pinned = -1;   // a trainer handle's snapshot, in force for the duration of its calls
int gs_deterministic() { return g_deterministic_pinned >= 0 ? g_deterministic_pinned : g_deterministic; }
void gs_pin_deterministic(int v) { g_deterministic_pinned = v; }

static CamParams make_cam(const float* view, const float* proj, const float* campos, float tanfovx, float tanfovy,
                          float scale_modifier, int W, int H) {
  CamParams cp;
  cp.view = view; cp.proj = proj; cp.campos = campos;
  cp.tanfovx = tanfovx; cp.tanfovy = tanfovy;
  cp.focal_x = W / (2.0f * tanfovx); cp.focal_y = H / (2.0f * tanfovy);
  cp.scale_modifier = scale_modifier;
  cp.scale_grad_factor = g_scale_grad_exact ? scale_modifier : 1.0f;
  cp.W = W; cp.H = H;
  cp.gx = (W + GS_TILE - 1) / GS_TILE; cp.gy = (H + GS_TILE - 1) / GS_TILE;
  return cp;
}

static inline uint32_t clamp_capacity(int64_t c) { return c <= 0 ? 0u : (c > 0x7fffffffLL ? 0x7fffffffu : (uint32_t)c); }

extern "C" {

int mi355gs_abi_version(void) { return MI355GS_ABI_VERSION; }

const char* mi355gs_error_string(int code) {
  switch (code) {
    case MI355GS_OK: return "ok";
    case MI355GS_EINVAL: return "invalid argument";
    case MI355GS_ELAUNCH: return "HIP launch or runtime failure";
    case MI355GS_EOVERFLOW: return "instance capacity exceeded";
    default: return "unknown error";
  }
}

size_t mi355gs_raster_geom_bytes(int P) { return GeomLayout(P).total; }
size_t mi355gs_raster_tiles_bytes(int W, int H) { return (W > 0 && H > 0) ? TilesLayout(W, H).total : 0; }
size_t mi355gs_raster_binning_bytes(int64_t n, int W, int H) {
  if (W <= 0 || H <= 0) return 0;
  return BinningLayout(n, TilesLayout(W, H).T).total;
}
size_t mi355gs_raster_grad_gate_offset(int P) { return gs_align((size_t)(P > 0 ? P : 1) * sizeof(GsGrad)); }
size_t mi355gs_raster_grad_scratch_bytes(int P) {
  return mi355gs_raster_grad_gate_offset(P) + 256 + (gs_deterministic() ? DetScratchLayout(P).total : 0);
}

int mi355gs_raster_forward_preprocess(void* stream_, int P, int D, int M, int W, int H, const float* means3D, const float* shs,
                                      const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                      const float* viewmatrix, const float* projmatrix, const float* campos, float tanfovx,
                                      float tanfovy, int prefiltered, int32_t* radii, void* geom, void* tiles,
                                      int32_t* num_rendered, uint8_t* visible, void* grad_scratch, int debug) {
  GS_RANGE();
  (void)prefiltered;  // as in the reference operator it only affects an internal consistency check
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || W <= 0 || H <= 0 || W > 65535 * GS_TILE || H > 65535 * GS_TILE || D < 0 || D > 3) return MI355GS_EINVAL;
  if (!geom || !tiles || !num_rendered || !viewmatrix || !projmatrix || !campos) return MI355GS_EINVAL;
  if (P > 0 && (!means3D || !opacities || !radii)) return MI355GS_EINVAL;
  if (P > 0 && ((shs == nullptr) == (colors_precomp == nullptr))) return MI355GS_EINVAL;
  if (P > 0 && shs && M < (D + 1) * (D + 1)) return MI355GS_EINVAL;
  if (shs_rest && (!shs || M < 2)) return MI355GS_EINVAL;
  if (P > 0 && !cov3D_precomp && (!scales || !rotations)) return MI355GS_EINVAL;
  const GeomLayout gl(P);
  const TilesLayout tl(W, H);
  char* g = (char*)geom;
  char* t = (char*)tiles;
  // The frame's accumulators are cleared by the projection kernel on its way (the one-call step hands over its own list): the
  // per-tile count + cursor words (adjacent), and — if the caller already holds the buffer its backward will use — the moment
  // records with the gate flags behind them.  No Gaussians, no kernel: memsets then.
  GsPrologue pro = g_fused.prologue;
  if (!g_fused.skip_memsets) {
    pro = GsPrologue();
    pro.tile_counters = (uint32_t*)(t + tl.count); pro.n_counters = (int)((tl.start - tl.count) / 4);
    if (grad_scratch) { pro.grad_records = (float4*)grad_scratch; pro.n_vec = mi355gs_raster_grad_gate_offset(P) / 16 + 2; }
    if (P <= 0) {
      if (hipMemsetAsync(t + tl.count, 0, tl.start - tl.count, stream) != hipSuccess) return MI355GS_ELAUNCH;
      if (grad_scratch && hipMemsetAsync(grad_scratch, 0, mi355gs_raster_grad_gate_offset(P) + 32, stream) != hipSuccess) return MI355GS_ELAUNCH;
    }
  }
  const CamParams cp = make_cam(viewmatrix, projmatrix, campos, tanfovx, tanfovy, scale_modifier, W, H);
  GS_KRANGE("preprocess_fwd");
  gs_launch_preprocess_fwd(stream, P, D, M, means3D, shs, shs_rest, colors_precomp, opacities, scales, rotations, cov3D_precomp, cp, radii,
                           (GsRec*)(g + gl.rec), (float*)(g + gl.cov3D), (uint2*)(g + gl.rect), (uint8_t*)(g + gl.clamped), (float*)(g + gl.depth), visible, pro);
  GS_CHECK_LAUNCH("preprocess_fwd");
  GS_KRANGE("count_tiles");
  gs_launch_count_tiles(stream, P, tl.T, tl.gx, (const uint2*)(g + gl.rect), (uint32_t*)(t + tl.count), (uint32_t*)(g + gl.bin_entries),
                        (uint32_t*)(g + gl.bin_n));
  GS_CHECK_LAUNCH("count_tiles");
  GS_KRANGE("scan_tiles");
  gs_launch_scan_tiles(stream, tl.T, (const uint32_t*)(t + tl.count), (uint32_t*)(t + tl.start), num_rendered,
                       (uint32_t*)(t + tl.order), (uint32_t*)(t + tl.meta), (uint32_t*)(t + tl.seg_first), (uint32_t*)(t + tl.part_first));
  GS_CHECK_LAUNCH("scan_tiles");
  return MI355GS_OK;
}

// Stage 2 of the forward in its two forms: `train` leaves the backward's work units, boundary records, hit masks and quadrant
// maxima in `binning` / `tiles`; render-only has a `binning` of keys + lists only and writes the image and the per-pixel state.
static int forward_stage2(void* stream_, int P, int W, int H, int64_t capacity, const float* bg, const void* geom, void* tiles,
                          void* binning, float* out_color, int debug, bool train) {
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || W <= 0 || H <= 0 || !geom || !tiles || !bg || !out_color) return MI355GS_EINVAL;
  const uint32_t cap = clamp_capacity(capacity);
  if (cap > 0 && !binning) return MI355GS_EINVAL;
  const GeomLayout gl(P);
  const TilesLayout tl(W, H);
  const BinningLayout bl(capacity, tl.T);   // (keys and list come first in it: the render-only buffer is its head)
  const char* g = (const char*)geom;
  char* t = (char*)tiles;
  char* b = (char*)binning;
  GS_KRANGE("binning");
  gs_launch_binning(stream, P, tl.T, tl.gx, (const float*)(g + gl.depth), (const uint2*)(g + gl.rect),
                    (const uint32_t*)(t + tl.start), (uint32_t*)(t + tl.cursor), (uint64_t*)(b + bl.keys),
                    (uint32_t*)(b + bl.list), cap, (const uint32_t*)(t + tl.order), (const uint32_t*)(g + gl.bin_entries),
                    (const uint32_t*)(g + gl.bin_n));
  GS_CHECK_LAUNCH("binning");
  {
    ProfScope prof(train ? 0 : 2, stream);
    GS_KRANGE("composite_fwd");
    gs_launch_composite_fwd(stream, tl.T, tl.gx, W, H, cap, (const uint32_t*)(t + tl.start), (const uint32_t*)(b + bl.list),
                            (const GsRec*)(g + gl.rec), bg, out_color, (float*)(t + tl.final_T), (uint32_t*)(t + tl.n_contrib),
                            (const uint32_t*)(t + tl.order), (const uint32_t*)(t + tl.seg_first), (const uint32_t*)(t + tl.part_first),
                            train ? (uint4*)(b + bl.unit_tile) : nullptr, train ? (float4*)(b + bl.bstate) : nullptr, bl.max_units,
                            (const uint32_t*)(t + tl.meta), train ? (unsigned long long*)(b + bl.hitmask) : nullptr, bl.max_chunks,
                            (uint32_t*)(t + tl.qmax), train ? g_prof.work_counters : nullptr, train);
  }
  GS_CHECK_LAUNCH("composite_fwd");
  return MI355GS_OK;
}

int mi355gs_raster_forward_render(void* stream, int P, int W, int H, int64_t capacity, const float* bg, const void* geom,
                                  void* tiles, void* binning, float* out_color, int debug) {
  GS_RANGE();
  return forward_stage2(stream, P, W, H, capacity, bg, geom, tiles, binning, out_color, debug, true);
}

size_t mi355gs_raster_binning_bytes_render_only(int64_t n, int W, int H) {
  if (W <= 0 || H <= 0) return 0;
  return BinningLayout(n, TilesLayout(W, H).T).unit_tile;   // keys + list: everything in front of the backward's tables
}

int mi355gs_raster_forward_render_only(void* stream, int P, int W, int H, int64_t capacity, const float* bg, const void* geom,
                                       void* tiles, void* binning, float* out_color, int debug) {
  GS_RANGE();
  return forward_stage2(stream, P, W, H, capacity, bg, geom, tiles, binning, out_color, debug, false);
}

int mi355gs_raster_backward(void* stream_, int P, int D, int M, int W, int H, const float* bg, const float* means3D,
                            const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities, const float* scales,
                            float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* campos, float tanfovx, float tanfovy, const void* geom,
                            void* tiles, const void* binning, int64_t capacity, const int32_t* radii, const float* out_color,
                            const float* dL_dpix, void* grad_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                            float* dL_dshs_rest, float* dL_dcolors, float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            int grad_scratch_is_clear, int debug) {
  GS_RANGE();
  (void)opacities; (void)colors_precomp;
  hipStream_t stream = (hipStream_t)stream_;
  if (P < 0 || W <= 0 || H <= 0 || D < 0 || D > 3) return MI355GS_EINVAL;
  if (!geom || !tiles || !dL_dpix || !out_color || !grad_scratch || !bg || !viewmatrix || !projmatrix || !campos) return MI355GS_EINVAL;
  if (P > 0 && (!means3D || !radii || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacities)) return MI355GS_EINVAL;
  const int use_shs = shs != nullptr, use_cov = cov3D_precomp != nullptr;
  if (P > 0 && use_shs && !dL_dshs) return MI355GS_EINVAL;
  if (shs_rest && !use_shs) return MI355GS_EINVAL;
  if (P > 0 && !use_shs && !dL_dcolors) return MI355GS_EINVAL;
  if (P > 0 && !use_cov && (!scales || !rotations || !dL_dscales || !dL_drotations)) return MI355GS_EINVAL;
  if (P > 0 && use_cov && !dL_dcov3D) return MI355GS_EINVAL;
  if (P == 0) return MI355GS_OK;
  const uint32_t cap = clamp_capacity(capacity);
  if (cap > 0 && !binning) return MI355GS_EINVAL;
  const GeomLayout gl(P);
  const TilesLayout tl(W, H);
  const BinningLayout bl(capacity, tl.T);
  const char* g = (const char*)geom;
  const char* t = (const char*)tiles;
  const char* b = (const char*)binning;
  GsGrad* grads = (GsGrad*)grad_scratch;
  // (with gate_tail the eight gate flags behind the records are cleared by the same memset)
  const size_t clear_bytes = g_fused.gate_tail ? mi355gs_raster_grad_gate_offset(P) + 8 * sizeof(float) : (size_t)P * sizeof(GsGrad);
  if (!g_fused.skip_memsets && !grad_scratch_is_clear && hipMemsetAsync(grads, 0, clear_bytes, stream) != hipSuccess) return MI355GS_ELAUNCH;
  if (cap > 0) {
    // deterministic mode: every instance's moments go to a row of their own (binning: det_rows / det_rowidx) and are summed per
    // Gaussian in rectangle order by k_det_gather, which writes every record — instead of meeting in float atomics
    const bool det = gs_deterministic() != 0;
    char* det_scratch = (char*)grad_scratch + mi355gs_raster_grad_gate_offset(P) + 256;
    uint32_t* rowidx = det ? (uint32_t*)(const_cast<char*>(b) + bl.det_rowidx) : nullptr;
    float* rows = det ? (float*)(const_cast<char*>(b) + bl.det_rows) : nullptr;
    if (det) {
      if (gs_launch_det_prepare(stream, P, tl.T, tl.gx, cap, (const uint32_t*)(t + tl.start), (const uint32_t*)(b + bl.list),
                                (const uint2*)(g + gl.rect), det_scratch, rowidx, rows) != 0) return MI355GS_ELAUNCH;
      GS_CHECK_LAUNCH("det_prepare");
    }
    {
      ProfScope prof(1, stream);
      gs_launch_composite_bwd(stream, tl.gx, W, H, cap, (const uint32_t*)(t + tl.start), (const uint32_t*)(b + bl.list),
                              (const GsRec*)(g + gl.rec), bg, (const float*)(t + tl.final_T), (const uint32_t*)(t + tl.n_contrib),
                              dL_dpix, grads, out_color, (const uint4*)(b + bl.unit_tile),
                              (const float4*)(b + bl.bstate), (const uint32_t*)(t + tl.meta), bl.max_units, bl.may_loop,
                              (const unsigned long long*)(b + bl.hitmask), bl.max_chunks, (const uint32_t*)(t + tl.qmax),
                              det ? nullptr : g_prof.work_counters, rowidx, rows);
    }
    GS_KRANGE("composite_bwd");
    if (det) gs_launch_det_gather(stream, P, cap, det_scratch, rows, grads);
    GS_CHECK_LAUNCH("composite_bwd");
  }
  const CamParams cp = make_cam(viewmatrix, projmatrix, campos, tanfovx, tanfovy, scale_modifier, W, H);
  GS_KRANGE("preprocess_bwd");
  gs_launch_preprocess_bwd(stream, P, D, M, means3D, shs, shs_rest, scales, rotations, use_shs, use_cov, cp, radii,
                           (const GsRec*)(g + gl.rec), (const float*)(g + gl.cov3D), (const uint8_t*)(g + gl.clamped), grads, dL_dmeans3D, dL_dmeans2D,
                           dL_dshs, dL_dshs_rest, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D,
                           (g_fused.gate && g_fused.gate_sh >= 0) ? g_fused.gate + g_fused.gate_sh : nullptr,
                           (g_fused.gate && g_fused.gate_sh_rest >= 0) ? g_fused.gate + g_fused.gate_sh_rest : nullptr);
  GS_CHECK_LAUNCH("preprocess_bwd");
  return MI355GS_OK;
}

int mi355gs_tune_deterministic(int on) {
  const int old = g_deterministic;
  if (on >= 0) g_deterministic = on ? 1 : 0;
  return old;
}

int mi355gs_tune_scale_grad(int mode) {
  const int old = g_scale_grad_exact;
  if (mode >= 0) g_scale_grad_exact = mode ? 1 : 0;
  return old;
}

int mi355gs_raster_frame_stats(void* stream_, int W, int H, const void* tiles, int64_t* stats) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (W <= 0 || H <= 0 || !tiles || !stats) return MI355GS_EINVAL;
  const TilesLayout tl(W, H);
  const char* t = (const char*)tiles;
  if (hipMemsetAsync(stats, 0, 2 * sizeof(int64_t), stream) != hipSuccess) return MI355GS_ELAUNCH;
  GS_KRANGE("frame_stats");
  gs_launch_frame_stats(stream, tl.T, tl.gx, W, H, (const uint32_t*)(t + tl.start), (const uint32_t*)(t + tl.n_contrib), stats);
  GS_CHECK_LAUNCH("frame_stats");
  return MI355GS_OK;
}

int mi355gs_profile_begin(void) {
  g_prof.on = true;
  for (int k = 0; k < PROF_KINDS; ++k) g_prof.used[k] = g_prof.seen[k] = 0;
  return MI355GS_OK;
}

int mi355gs_profile_set_period(int every) {
  const int old = g_prof.period;
  if (every > 0) g_prof.period = every;
  return old;
}

int mi355gs_profile_work_counters(void* counters) {
  g_prof.work_counters = (unsigned long long*)counters;
  return MI355GS_OK;
}

int mi355gs_profile_read(int kind, double* total_ms, int* launches) {
  if (kind < 0 || kind >= PROF_KINDS || !total_ms || !launches) return MI355GS_EINVAL;
  double tot = 0.0;
  for (int i = 0; i < g_prof.used[kind]; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[kind][i][1]) != hipSuccess) return MI355GS_ELAUNCH;
    if (hipEventElapsedTime(&ms, g_prof.ev[kind][i][0], g_prof.ev[kind][i][1]) != hipSuccess) return MI355GS_ELAUNCH;
    tot += ms;
  }
  *total_ms = tot;
  *launches = g_prof.used[kind];
  return MI355GS_OK;
}

int mi355gs_profile_end(void) {
  g_prof.on = false;
  return MI355GS_OK;
}

int mi355gs_profile_ranges(int on) {
  if (on == 1) {
    if (!roctx_open()) return MI355GS_EINVAL;
    g_ranges_on = true;
  } else if (on == 0) {
    g_ranges_on = false;
  }
  return (int)(g_roctx.pushed & 0x7fffffffull);
}

int mi355gs_raster_mark_visible(void* stream_, int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                uint8_t* present) {
  GS_RANGE();
  (void)projmatrix;
  const int debug = 0;
  void* stream = stream_;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return MI355GS_EINVAL;
  GS_KRANGE("mark_visible");
  gs_launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
  GS_CHECK_LAUNCH("mark_visible");
  return MI355GS_OK;
}

}  // extern "C"
