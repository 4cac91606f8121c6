// Per-tile alpha compositing, forward and backward (A.4/A.5).
//
// Workgroup = one 16x16 tile = 4 wave64; each wave owns an 8x8 pixel quadrant (lane -> pixel inside
// the quadrant), so a whole wave can skip a Gaussian whose alpha>=1/255 box misses its quadrant:
// the batch of 256 staged records is tested 64 at a time (one record per lane), the hits are
// collected with a 64-bit ballot, and the wave then walks the set bits of that (scalar) mask —
// record addresses in LDS are wave-uniform, so the loads are broadcasts and the loop control is
// pure SALU.  Records are gathered from the 48-byte per-Gaussian array (L2 / Infinity-Cache
// resident) through the depth-sorted per-tile index list.
#include "common.h"

// ---- measurement build only (make probe -> lib/libmi355gs_probe.so, loaded by tools/probe_composite.py, never by the
// package): per-wave clocks and counts from inside the composite kernels.  The product library is compiled without it.
#ifdef GS_PROBE
__device__ unsigned long long* g_probe_buf = nullptr;
__device__ unsigned int g_probe_cap = 0;
extern "C" int mi355gs_probe_set(void* buf, unsigned int capacity_rows) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_probe_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_probe_cap), &capacity_rows, sizeof(capacity_rows)) == hipSuccess ? 0 : -1;
}
#define GS_PROBE_ROW 8
#define GS_PROBE_CLOCK() ((unsigned long long)wall_clock64())
#define GS_PROBE_STORE(row, c0, c1, c2, c3, c4, c5, c6, c7)                                          \
  do {                                                                                                 \
    if (g_probe_buf && (row) < g_probe_cap && (threadIdx.x & 63) == 0) {                               \
      unsigned long long* p_ = g_probe_buf + (size_t)(row) * GS_PROBE_ROW;                            \
      p_[0] = (c0); p_[1] = (c1); p_[2] = (c2); p_[3] = (c3); p_[4] = (c4); p_[5] = (c5); p_[6] = (c6); p_[7] = (c7); \
    }                                                                                                  \
  } while (0)
#else
#define GS_PROBE_CLOCK() 0ull
#define GS_PROBE_STORE(...) do {} while (0)
#endif

// A/B build switch (tools/build_variant.sh precise -DGS_PRECISE_MATH=1): the library's exp2f and an IEEE division in place of
// v_exp_f32 / v_rcp_f32 in the alpha and transmittance arithmetic of both kernels — what the hardware approximations cost in
// accuracy against the float64 oracle is measured with it (profiles/EXPERIMENTS.md, round 5); the shipped build has it off.
#ifndef GS_PRECISE_MATH
#define GS_PRECISE_MATH 0
#endif
__device__ __forceinline__ float gs_exp2(float x) { return GS_PRECISE_MATH ? exp2f(x) : __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float gs_rcp(float x) { return GS_PRECISE_MATH ? 1.0f / x : __builtin_amdgcn_rcpf(x); }

namespace {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

struct Quad {
  int px, py;          // this lane's pixel
  gs_v2f f;            // as float (x, y)
  float x0, x1, y0, y1;  // the wave's 8x8 pixel box (inclusive, float)
  bool inside;
};

__device__ __forceinline__ Quad make_quad_xy(int tx, int ty, int W, int H) {
  const int wave = (threadIdx.x >> 6) & 3, lane = threadIdx.x & 63;   // (the forward runs FWD_TILES tiles of four waves per workgroup)
  const int bx = tx * GS_TILE + (wave & 1) * 8, by = ty * GS_TILE + (wave >> 1) * 8;
  Quad q;
  q.px = bx + (lane & 7);
  q.py = by + (lane >> 3);
  q.f = gs_v2f{(float)q.px, (float)q.py};
  q.x0 = (float)bx; q.x1 = (float)(bx + 7); q.y0 = (float)by; q.y1 = (float)(by + 7);
  q.inside = q.px < W && q.py < H;
  return q;
}
__device__ __forceinline__ Quad make_quad(int tile, int gx, int W, int H) { return make_quad_xy(tile % gx, tile / gx, W, H); }

__device__ __forceinline__ bool box_hit(const float4& q0, const Quad& q) {
  return (q0.x + q0.z >= q.x0) && (q0.x - q0.z <= q.x1) && (q0.y + q0.w >= q.y0) && (q0.y - q0.w <= q.y1);
}

// Exact test "can any pixel of the wave's 8x8 box reach alpha >= 1/255 for this Gaussian?":
// the box test above, then the maximum of the (concave) exponent over the box — 0 if the centre is
// inside, otherwise the best of the four edges (a 1-D quadratic each, maximiser clamped to the edge).
// Runs once per (wave, staged record) with 64 records per instruction, so its ~45 ops cost < 1 op per
// hit, and it removes the corner quadrants an axis-aligned box lets through (~15 % of the hits).
// Conservative: 1 % + 1e-3 slack on the threshold; skipped (box only) when the conic is ill-conditioned.
__device__ __forceinline__ bool quad_hit(const float4& q0, const float4& q1, const Quad& q) {
  if (!box_hit(q0, q)) return false;
  if (q0.z > 1e29f) return true;
  const float gx = q0.x, gy = q0.y;
  if (gx >= q.x0 && gx <= q.x1 && gy >= q.y0 && gy <= q.y1) return true;
  // q1 holds the exp2-scaled conic: e(dx,dy) = A dx^2 + C dy^2 + B dx dy (A, C < 0) is log2 of the Gaussian falloff
  const float A = q1.x, C = q1.y, B = q1.z;
  const float tau = __log2f(255.0f * q1.w) * 1.01f + 1e-3f;
  const float inv_2A = __builtin_amdgcn_rcpf(2.0f * A), inv_2C = __builtin_amdgcn_rcpf(2.0f * C);
  const float dx_lo = gx - q.x1, dx_hi = gx - q.x0, dy_lo = gy - q.y1, dy_hi = gy - q.y0;
  float best = -3.0e38f;
  {
    const float dx = dx_hi, dy = fminf(dy_hi, fmaxf(dy_lo, -B * dx * inv_2C));
    best = fmaxf(best, A * dx * dx + C * dy * dy + B * dx * dy);
  }
  {
    const float dx = dx_lo, dy = fminf(dy_hi, fmaxf(dy_lo, -B * dx * inv_2C));
    best = fmaxf(best, A * dx * dx + C * dy * dy + B * dx * dy);
  }
  {
    const float dy = dy_hi, dx = fminf(dx_hi, fmaxf(dx_lo, -B * dy * inv_2A));
    best = fmaxf(best, A * dx * dx + C * dy * dy + B * dx * dy);
  }
  {
    const float dy = dy_lo, dx = fminf(dx_hi, fmaxf(dx_lo, -B * dy * inv_2A));
    best = fmaxf(best, A * dx * dx + C * dy * dy + B * dx * dy);
  }
  return best >= -tau;
}

// log2 of the Gaussian falloff at offset (dx, dy): (d * (A, C)) * d summed, then fma(B dx, dy, sum) — the operation order of
// the forward's packed form, with FMA contraction of the two squares switched off.  The backward must reproduce the
// forward's alpha bit for bit, or a pair within rounding of alpha = 1/255 is blended by one pass and skipped by the other.
__device__ __forceinline__ float gs_power2(float dx, float dy, float A, float C, float B) {
#pragma clang fp contract(off)
  const float sx = (dx * A) * dx, sy = (dy * C) * dy;
  const float sum = sx + sy;
  return fmaf(B * dx, dy, sum);
}

// ------------------------------------------------------------------------------------------------
// K6 forward.  Workgroup = one 16x16 tile (heaviest tiles first), wave = one 8x8 pixel quadrant of it — and the four waves
// never meet: each stages its own 64-record groups in wave-private LDS (lane i <- instance base + i, gathered through the
// sorted list; the next group's records are requested before the current one is walked), tests the 64 staged records against
// its quadrant at once (one per lane: box, then the exact maximum of the exponent over the box), collects the hits with a
// ballot, and walks the set bits of that scalar mask: record addresses are wave-uniform (LDS broadcast reads), loop control is
// SALU.  The walk takes FWD_GROUP hits per step: their alphas side by side (record reads, exponent, v_exp_f32, threshold tests:
// independent), then the short transmittance-recurrence steps; a short group is padded with alpha = 0, a no-op for every
// state variable (a live T >= 1e-4 stays, a finished one stays finished; w == 0 leaves colour and `last` alone).
//
// Round 1/2 ran this stage as persistent workgroups over CU-balanced bins with cooperative 512-record batches (two workgroup
// barriers per batch).  Measured on MI355X, same frames: 84.7 -> 81.8 us at C3 (512^2), 623 -> 595 us at C4 (1080p) for this
// form, bit-identical images — a wave that finishes early no longer waits for its tile's slowest quadrant, 59 VGPRs keep eight
// waves per SIMD, and neither the scheduler's atomics nor its bookkeeping in k_scan_tiles exist any more.
// ------------------------------------------------------------------------------------------------
#ifndef GS_FWD_GROUP
#define GS_FWD_GROUP 2
#endif
constexpr int FWD_GROUP = GS_FWD_GROUP;   // hits per walk step (A/B build switch)

// FWD_TILES tiles per workgroup, dealt serpentine from the heaviest-first order (positions p, 2Q - 1 - p, 2Q + p, 4Q - 1 - p):
// sixteen independent waves, four per SIMD — quadrant q of all four tiles.  A 512^2 frame is one resident round of 4096 waves,
// nothing rebalances it, and the kernel lasts as long as its busiest CU (probe: hits per CU up to 1.17x the mean, correlation
// with the finish time 0.92).  Which CU a workgroup lands on cannot be chosen, but what a workgroup weighs can: a heavy, a light
// and two middling tiles weigh about the same in every workgroup, and with 256 workgroups for 256 CUs a CU's load is one
// workgroup's.  Measured at C3 (profiles/r03_ab_fwd_tiles_per_workgroup_kernel_avg.txt): 1 tile per workgroup 77.15 us, 2: 75.1, 4: 74.1.
// A frame with more tiles than that runs in several rounds, which the dispatcher balances by itself, and big workgroups only
// coarsen its grain (C4, 8160 tiles: 581 us with one tile per workgroup, 606 with four): the launcher picks FWD_TILES per frame.
#ifndef GS_FWD_TILES
#define GS_FWD_TILES 0   // 0: by the frame's tile count (gs_launch_composite_fwd); 1, 2, 4: forced (A/B builds)
#endif
// COUNT: the measurement instantiation (mi355gs_profile_work_counters, like the backward's): the same kernel also adds up, per
// wave, its staged groups, hits, walk steps and the lanes of a hit that hold a blendable (pixel, Gaussian) pair — the inputs of
// the forward's VALU-issue model in bench.py (counters[8 ..]).  The shipped launches use COUNT = false: no counters exist.
// TRAIN = false: the render-only instantiation (mi355gs_raster_forward_render_only: every no-grad render — evaluation, the FPS
// loop, reference render.py:87,137,177 under torch.no_grad()).  Nothing is left for a backward that will not come: no unit
// table, no 16 B/pixel boundary record per 64-instance unit, no hit masks, no quadrant maxima (51 MB of stores per 512^2 frame
// at C3 against 5 MB of image and per-pixel state, profiles/r04_pmc_c3_WRITE_SIZE.csv), and none of their address arithmetic.
// The image, final_T and n_contrib are bit-identical to the training instantiation's: the walk is the same code.
template <int FWD_TILES, bool COUNT = false, bool TRAIN = true>
__global__ __launch_bounds__(256 * FWD_TILES) void k_composite_fwd(int T, int gx, int W, int H, uint32_t capacity, const uint32_t* __restrict__ tile_start,
                                                        const uint32_t* __restrict__ list, const GsRec* __restrict__ recs,
                                                        const float* __restrict__ bg, float* __restrict__ out_color,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                        const uint32_t* __restrict__ order, const uint32_t* __restrict__ seg_first,
                                                        const uint32_t* __restrict__ part_first, uint4* __restrict__ unit_tile,
                                                        float4* __restrict__ bstate, uint32_t max_units,
                                                        const uint32_t* __restrict__ meta, unsigned long long* __restrict__ hitmask,
                                                        uint32_t max_chunks, uint32_t* __restrict__ qmax,
                                                        unsigned long long* __restrict__ counters) {
  [[maybe_unused]] unsigned long long c_groups = 0, c_hits = 0, c_steps = 0, c_valid = 0, c_blended = 0;
  // The wave's staged group: 64 records of three float4 each, record-major (48 B apiece), at an address the SCALAR unit knows —
  // the wave index is read into an SGPR and the walk's record index is scalar already, so a hit's three broadcast reads take
  // ONE address register filled by a v_mov from an SGPR plus immediate offsets.  (Three separate arrays indexed through the
  // per-thread wave index cost three v_add_u32 per hit: 12 of the ~78 VALU issue cycles of a hit.)  Lane i's three 16-byte
  // stores at a 48-byte pitch are bank-conflict-free (12 i mod 64 enumerates sixteen disjoint groups of four banks).
  __shared__ float4 s_rec[4 * FWD_TILES][GS_SEG][3];
  const int tid = threadIdx.x & 255, lane = tid & 63;   // tid: the thread's index inside its TILE (pixel state, boundary records)
  const int wave_wg = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 .. 4 FWD_TILES - 1
  const int wave = wave_wg & 3;                                               // the quadrant
  // position in the heaviest-first order: tile group j of workgroup p takes p, 2Q - 1 - p, 2Q + p, 4Q - 1 - p (Q workgroups): a
  // serpentine deal, every position exactly once
  const int nwg = (T + FWD_TILES - 1) / FWD_TILES, j = wave_wg >> 2;
  const int pos = (j & 1) ? (j + 1) * nwg - 1 - (int)blockIdx.x : j * nwg + (int)blockIdx.x;
  if (pos >= T) return;   // (tile count not a multiple of FWD_TILES; no workgroup barrier below)
  const int tile = (int)order[pos];
  const uint32_t seg_len = TRAIN ? meta[2] * GS_SEG : 0u;    // instances per backward unit of this frame (k_scan_tiles)
  float4 (*__restrict__ recl)[3] = s_rec[wave_wg];
  const Quad q = make_quad(tile, gx, W, H);
  const uint32_t start = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  [[maybe_unused]] const unsigned long long pr_t0 = GS_PROBE_CLOCK();
  [[maybe_unused]] unsigned long long pr_hits = 0, pr_groups = 0, pr_walk = 0;
  // Backward units of this tile (segments of seg_len instances, see common.h): publish them, and leave every pixel's
  // (transmittance after the last blended Gaussian, accumulated colour) at each segment boundary for the backward.
  const uint32_t seg0 = TRAIN ? seg_first[tile] : 0u, nseg = TRAIN ? seg_first[tile + 1] - seg0 : 0u;
  // The table is written in the backward's LAUNCH order: every full-length unit of the frame first (tile by tile), then the
  // tiles' short last units.  The hardware hands out workgroups in index order, so the units that start last are the short ones
  // and the kernel's tail — the stretch where the SIMDs run out of waves — is made of half-length work.  A unit's boundary
  // record stays at its slot seg_first[tile] + segment; the entry carries it.
  if constexpr (TRAIN) {
    const uint32_t p0 = part_first[tile], n_full = nseg - (part_first[tile + 1] - p0);
    const uint32_t full0 = seg0 - p0, all_full = meta[1] - meta[3];
    for (uint32_t sg = tid; sg < nseg; sg += 256) {
      const uint32_t pos = sg < n_full ? full0 + sg : all_full + p0;
      if (pos < max_units)  // (tile x | tile y << 16, segment, slot): the backward needs no division to place itself
        unit_tile[pos] = make_uint4((uint32_t)(tile % gx) | ((uint32_t)(tile / gx) << 16), sg, seg0 + sg, 0u);
    }
  }
  [[maybe_unused]] uint32_t next_boundary = 0;  // boundaries [0, next_boundary) of this tile have been stored by this wave

  // Tr: live transmittance while the pixel is still blending; once it is finished (T < 1e-4 would be reached) it holds MINUS
  // the transmittance after the last blended Gaussian, so |Tr| is always the value the reference stores as final_T and the
  // sign is the "done" flag (outside the image: -1 from the start)
  float Tr = q.inside ? 1.0f : -1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;
  // The staged record of this lane as three native 16-byte vectors (a float4 local that is assigned under a condition ends up
  // in scratch memory, and the store that puts it there waits for the load it was meant to overlap).  Gathering a record is
  // two dependent round trips (sorted list -> record); both are taken off the critical path: while group g is walked, the
  // records of group g + 1 and the list entries of group g + 2 are in flight.  Addresses are clamped, so no load sits in a branch.
  typedef float v4f __attribute__((vector_size(16)));   // (vector_size, not ext_vector_type: g++ builds these sources for the emulator)
  auto list_at = [&](uint32_t j) { return list[min(j, end - 1u)]; };
  auto fetch = [&](uint32_t id, v4f& a, v4f& b, v4f& c) {
    const v4f* r = reinterpret_cast<const v4f*>(recs + id);
    a = r[0]; b = r[1]; c = r[2];
  };
  v4f r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0;
  uint32_t id_next = 0;
  if (start < end) {
    fetch(list_at(start + (uint32_t)lane), r0, r1, r2);
    id_next = list_at(start + GS_SEG + (uint32_t)lane);
  }
  for (uint32_t base = start; base < end; base += GS_SEG) {
    const int cnt = (int)min((uint32_t)GS_SEG, end - base);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the previous group's LDS reads are done (only this wave reads its staging area)
    *reinterpret_cast<v4f*>(&recl[lane][0]) = r0; *reinterpret_cast<v4f*>(&recl[lane][1]) = r1; *reinterpret_cast<v4f*>(&recl[lane][2]) = r2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool hit = lane < cnt && quad_hit(make_float4(r0[0], r0[1], r0[2], r0[3]), make_float4(r1[0], r1[1], r1[2], r1[3]), q);
    unsigned long long mask = __ballot(hit);
    if constexpr (TRAIN) {
      // the backward's cull of this chunk for this quadrant, done: chunk index = first unit of the tile * chunks per unit + group
      const uint32_t chunk = seg0 * meta[2] + (base - start) / GS_SEG;
      if (lane == 0 && chunk < max_chunks) hitmask[(size_t)chunk * 4 + wave] = mask;
    }
    if (base + GS_SEG < end) {
      fetch(id_next, r0, r1, r2);
      id_next = list_at(base + 2 * GS_SEG + (uint32_t)lane);
    }
    // the boundary in front of this group (the state after `base - start` instances) is stored here, not behind the walk
    // that produced it: the wait for the next group's records at the top of the loop also covers every older memory
    // operation, and a store issued just before it would be waited for at its full latency
    if constexpr (TRAIN) {
      if (base != start && (base - start) % seg_len == 0u) {
        next_boundary = (base - start) / seg_len;
        if (seg0 + next_boundary - 1u < max_units) bstate[(size_t)(seg0 + next_boundary - 1u) * 256 + tid] = make_float4(fabsf(Tr), C0, C1, C2);
      }
    }
#ifdef GS_PROBE
    pr_hits += __popcll(mask); pr_groups += 1;
    const unsigned long long pr_k0 = GS_PROBE_CLOCK();
#endif
    if constexpr (COUNT) { c_groups += 1; c_hits += __popcll(mask); }
    while (mask) {
      if constexpr (COUNT) c_steps += 1;
      int idx[FWD_GROUP];
      bool live[FWD_GROUP];
#pragma unroll
      for (int u = 0; u < FWD_GROUP; ++u) {
        idx[u] = mask ? __ffsll(mask) - 1 : (u ? idx[u - 1] : 0);
        live[u] = mask != 0;   // wave-uniform: folds into the lane masks below as scalar logic
        mask &= mask - 1;      // 0 stays 0
      }
      float al[FWD_GROUP];
      bool valid[FWD_GROUP];
      float4 col[FWD_GROUP];
#pragma unroll
      for (int u = 0; u < FWD_GROUP; ++u) {
        const float4 a0 = recl[idx[u]][0], a1 = recl[idx[u]][1];
        col[u] = recl[idx[u]][2];
        const gs_v2f d = gs_v2f{a0.x, a0.y} - q.f;
        const gs_v2f sq = (d * gs_v2f{a1.x, a1.y}) * d;                     // packed: (A dx^2, C dy^2)
        const float power2 = fmaf(a1.z * d[0], d[1], sq[0] + sq[1]);        // log2 of the falloff
        const float alpha = fminf(0.99f, a1.w * gs_exp2(power2));
        // a skipped Gaussian is a transparent one (and so is the padding of a short group)
        valid[u] = live[u] && power2 <= 0.0f && alpha >= ALPHA_MIN;
        al[u] = valid[u] ? alpha : 0.0f;
        if constexpr (COUNT) c_valid += __popcll(__ballot(valid[u]));
      }
#pragma unroll
      for (int u = 0; u < FWD_GROUP; ++u) {
        // A finished pixel carries its final transmittance NEGATED (outside the image: -1): w0 and test_T are then
        // negative, so `stop` (and nothing else) also covers "already done", and one select updates the state.
        const float w0 = al[u] * Tr;
        const float test_T = Tr - w0;  // T (1 - alpha)
        const bool stop = test_T < T_MIN;
        const float w = stop ? 0.0f : w0;
        C0 += col[u].x * w; C1 += col[u].y * w; C2 += col[u].z * w;
        // blended <=> alpha passed the tests and this is not the stopping Gaussian (w > 0 exactly then): mask logic
        // on the two compare results instead of a third compare
        // (carrying this position in the staged record's unused depth slot — one broadcast read wider, no scalar-to-vector move
        // per hit — measured slower: 75.5 -> 78.1 us at C3; three or four hits per walk step: 78.2 / 81.7 us)
        last = (valid[u] && !stop) ? (base - start) + (uint32_t)idx[u] + 1u : last;
        if constexpr (COUNT) c_blended += __popcll(__ballot(valid[u] && !stop));
        Tr = stop ? -fabsf(Tr) : test_T;
      }
    }
#ifdef GS_PROBE
    pr_walk += GS_PROBE_CLOCK() - pr_k0;
#endif
    if (__all(Tr < 0.0f)) break;
  }
  // boundaries this wave never reached (all its pixels were finished, or the tile ended): the state no longer changes
  const float Tfin = fabsf(Tr);
  if constexpr (TRAIN)
    for (uint32_t sg = next_boundary; sg + 1u < nseg; ++sg)
      if (seg0 + sg < max_units) bstate[(size_t)(seg0 + sg) * 256 + tid] = make_float4(Tfin, C0, C1, C2);
  if constexpr (TRAIN) {
    // the quadrant's largest contributor count: every backward unit of the tile needs it, and computed there it is a wave
    // reduction per unit and quadrant (4 x 11 k per C3 frame) instead of one per forward wave
    const uint32_t wmax = gs_wave_max_u32(last);
    if (lane == 0) qmax[(size_t)tile * 4 + wave] = wmax;
  }
  if (q.inside) {
    const size_t pix = (size_t)q.py * W + q.px, plane = (size_t)W * H;
    final_T[pix] = Tfin;
    n_contrib[pix] = last;
    out_color[pix] = C0 + Tfin * bg[0];
    out_color[plane + pix] = C1 + Tfin * bg[1];
    out_color[2 * plane + pix] = C2 + Tfin * bg[2];
  }
  if constexpr (COUNT) {
    if (lane == 0 && counters) {
      atomicAdd(counters + 8, c_groups); atomicAdd(counters + 9, c_hits); atomicAdd(counters + 10, c_steps);
      atomicAdd(counters + 11, c_valid); atomicAdd(counters + 12, c_blended); atomicAdd(counters + 13, 1ull);
    }
  }
  GS_PROBE_STORE((uint32_t)tile * 4u + (uint32_t)wave, pr_t0, GS_PROBE_CLOCK(), pr_hits, 0ull, pr_groups, (unsigned long long)(end - start),
                 (unsigned long long)gs_physical_cu(), pr_walk);
}

// ------------------------------------------------------------------------------------------------
// K7 backward: replay each tile back to front.  Per (pixel, Gaussian) the nine screen-space
// gradient terms are reduced across the wave's 64 pixels in registers (butterfly shuffles) and
// leave the wave as ONE set of float atomics per Gaussian per wave (the reference operator issues
// them per pixel).  Gaussians whose box misses the wave's quadrant, or that lie behind every
// pixel's last contributor, are skipped wave-wide.
// ------------------------------------------------------------------------------------------------
// One WAVE = one UNIT: segment `seg` (GS_SEG instances) of one tile's list; a workgroup is BW_UNITS independent waves (one, below).
// The state a back-to-front replay would carry into the segment comes from the forward's boundary record instead: T is the
// forward's own product, and the colour behind is dL/dC . (final colour - colour accumulated in front of the boundary).
//
// A lane owns FOUR pixels, one in each 8x8 quadrant of the tile (same lane -> (x, y) offset inside every quadrant), so
//  * culling stays wave-uniform per quadrant (four ballots, one scalar mask walk over their union), and
//  * the nine moments of a Gaussian are first accumulated in registers over the lane's pixels (plain FMAs) and cross the
//    lanes ONCE per (Gaussian, tile) instead of once per (Gaussian, quadrant) — a Gaussian touches 1.5-2 quadrants of a tile
//    on the benchmark scenes, and round 1 paid the reduction (and nine atomics) for every one of them.  The reduction itself
//    goes through LDS since round 3 (BW_REDUCE_LDS below: ~50 VALU issue cycles); rounds 1-2 transposed it in registers
//    (8 v_permlane*_swap at 8 cycles + 9 DPP / select ops: ~125 issue cycles, more than a quadrant's pixel math).
constexpr uint32_t BW_XCD_RUN = 64;   // consecutive units sent to the same XCD (C3, FETCH_SIZE per launch: none 90 MB, 16: 56, 32: 50, 64: 47; time 124.6 / 122.1 / 122.0 / 122.7 us)
#ifndef GS_BW_MFMA
#define GS_BW_MFMA 0
#endif
#ifndef GS_BW_LDS
#define GS_BW_LDS 1   // shipped: measured on MI355X at C3 124.5 -> 109.4 us against the register-transposed reduction (GS_BW_LDS=0),
#endif                // 182 us with the cross-row half on the matrix pipe (GS_BW_MFMA=1); profiles/r03_ab_bwd_*
constexpr bool BW_REDUCE_MFMA = GS_BW_MFMA != 0;   // A/B build switches of the reduction in replay_one (tools/build_variant.sh)
constexpr bool BW_REDUCE_LDS = GS_BW_LDS != 0;
// BW_REDUCE_LDS: the nine per-lane moments cross the wave through LDS: nine moment-major rows of 64 floats (pitch RED_PITCH:
// 64 + 4, so that the 16-float quarters four neighbouring readers take start in different banks), written with nine
// ds_write_b32 (lane = column) and read back by 36 lanes — moment = lane / 4, quarter = lane % 4 — as four ds_read_b128 each;
// 15 adds, two quad_perm steps and lanes 0, 4, .., 32 hold the nine sums.  The LDS crossbar moves the data on its own pipe;
// the VALU is left with the additions (~40 issue cycles instead of ~128 for the register-transposed form).
constexpr int RED_PITCH = 68;
// GS_BW_HALF: every 64-instance unit of the one-chunk instantiation is replayed by TWO waves that never meet: the BACK half of
// the unit (instances 32..63) back to front from the unit's far boundary record, exactly as before, and the FRONT half
// (instances 0..31) FRONT TO BACK from the near boundary record (= the previous unit's far record; T = 1, C = 0 for a tile's
// first unit).  The forward already leaves both records (one per 64 instances): nothing changes on its side.  Front to back,
// the colour behind Gaussian k is dL/dC . (C_out - C accumulated through k) — the same running scalar, walked the other way —
// and the transmittance is the forward's own product T (1 - alpha), no division.  Twice the waves, half the length: the
// backward of a 512^2 frame is ~1.84 resident rounds of 64-instance units whose last third runs below four waves per SIMD
// (tools/probe_bwd.py), and the tail is as long as a unit.
#ifndef GS_BW_HALF
#define GS_BW_HALF 0
#endif
constexpr int BW_UNITS = 1;  // units (waves) per workgroup: single-wave workgroups give the dispatcher the finest grain (129.4 -> 128.2 us at C3 against 4)

// CHUNKS: 64-instance chunks per unit (1, or 0 = the frame's own value from meta[2] for the longer units of big frames; the
// one-chunk instantiation keeps 74 VGPRs / six waves per SIMD, the loop over chunks costs 15 more)
// COUNT: the measurement instantiation (mi355gs_profile_work_counters): the same kernel also adds up, per wave, how many
// (Gaussian, tile) steps it ran, how many quadrant bodies, and how many lanes of those bodies were valid pixels — the inputs
// of the VALU-issue model bench.py reports next to the HBM roofline.  The shipped launches use COUNT = false: no counters exist.
#ifdef GS_BW_MIN_WAVES
#define GS_BW_BOUNDS __launch_bounds__(64 * BW_UNITS, GS_BW_MIN_WAVES)   // A/B build switch: force a register budget
#else
#define GS_BW_BOUNDS __launch_bounds__(64 * BW_UNITS)
#endif
// DET: the deterministic instantiation (mi355gs_tune_deterministic): the staged record carries the instance's ROW index instead of
// the Gaussian's, and the nine sums of a step are stored to that row (every (Gaussian, tile) instance is replayed by exactly one
// unit, once) instead of being added to the Gaussian's record with float atomics; k_det_gather then sums a Gaussian's rows in
// the order of its tile rectangle.  Same arithmetic up to that final order of additions — which no longer depends on timing.
template <int CHUNKS, bool COUNT = false, bool DET = false>
__global__ GS_BW_BOUNDS void k_composite_bwd(int gx, int W, int H, uint32_t capacity,
                                                        const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ list,
                                                        const GsRec* __restrict__ recs, const float* __restrict__ bg,
                                                        const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix, GsGrad* __restrict__ grads,
                                                        const float* __restrict__ out_color,
                                                        const uint4* __restrict__ unit_tile, const float4* __restrict__ bstate,
                                                        const uint32_t* __restrict__ meta, uint32_t max_units,
                                                        unsigned long long* __restrict__ counters,
                                                        const unsigned long long* __restrict__ hitmask, uint32_t max_chunks,
                                                        const uint32_t* __restrict__ qmax, const uint32_t* __restrict__ det_rowidx,
                                                        float* __restrict__ det_rows) {
  static_assert(!DET || BW_REDUCE_LDS, "the deterministic instantiation is written for the LDS reduction");
  __shared__ float4 s_rec[BW_UNITS][GS_SEG][3];   // the staged chunk, record-major, at a scalar address (see k_composite_fwd)
  __shared__ __attribute__((aligned(16))) float s_red[BW_UNITS][BW_REDUCE_LDS ? 9 * RED_PITCH : 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = BW_UNITS == 1 ? 0 : (int)__builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
  // Workgroup p runs on XCD p mod 8 (round-robin dispatch), each XCD behind its own L2.  Consecutive units are mostly units of
  // one tile and re-read the same 8 KiB of per-pixel state, so they should share an L2: inside every block of 8 * BW_XCD_RUN
  // launch positions the index is transposed, and units RUN b .. RUN b + RUN - 1 of the block all land on XCD b.
  static_assert(BW_UNITS == 1, "the XCD transposition below is written for one unit per workgroup");
  constexpr bool HALVES = GS_BW_HALF != 0 && CHUNKS == 1;
  constexpr uint32_t PER_UNIT = HALVES ? 2u : 1u;   // launch positions per unit; the two halves of a unit are neighbours on one XCD
  const uint32_t pos = blockIdx.x, in_block = pos % (8u * BW_XCD_RUN * PER_UNIT);
  const uint32_t hu = pos - in_block + (in_block & 7u) * (BW_XCD_RUN * PER_UNIT) + (in_block >> 3);
  const uint32_t unit = hu / PER_UNIT;
  const bool fwd_dir = HALVES && (hu & 1u) == 0u;   // wave-uniform: this wave replays the unit's front half, front to back
  if (unit >= min(meta[1], max_units)) return;  // wave-uniform; no workgroup barrier below
  // The one-chunk instantiation is launched when the CAPACITY cannot need longer units; a frame that overflowed its capacity
  // may still have been laid out in longer ones (k_scan_tiles decides from the true count).  Such a frame is discarded by its
  // caller anyway: leave its gradients zero instead of replaying it with the wrong unit length.
  if (CHUNKS != 0 && meta[2] != (uint32_t)CHUNKS) return;
  [[maybe_unused]] const unsigned long long pr_t0 = GS_PROBE_CLOCK();
  [[maybe_unused]] unsigned long long pr_steps = 0, pr_t1 = 0;
  // Everything the prologue needs is requested in TWO rounds of loads instead of five dependent ones (placement -> tile range
  // -> pixel state -> wave maxima / branch -> boundary record): the table entry names the unit's tile, segment and the slot of
  // its boundary record, which is requested before the unit knows whether it will need it (the deepest unit of a tile does
  // not: 1 in ~11 at C3), together with the pixel state — including the frame's
  // colour, used only with the record — follows as soon as the tile is known, from clamped addresses so that no load sits
  // in a branch.  GS_PIN4 (an empty asm that "uses" the values) keeps the compiler from sinking the loads below the early
  // exits; it comes after the last load has been issued, where the first consumer would wait anyway.
#define GS_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
  const uint4 entry = unit_tile[unit];   // `unit` is the position in launch order (full-length units first, short ones last)
  const uint32_t where = __builtin_amdgcn_readfirstlane(entry.x), seg = __builtin_amdgcn_readfirstlane(entry.y);  // uniform: scalar
  const uint32_t slot = __builtin_amdgcn_readfirstlane(entry.z);
  if (slot >= max_units) return;   // (a frame that overflowed its buffers)
  float4 brec[4];
  const uint32_t bslot = (fwd_dir && seg > 0u) ? slot - 1u : slot;   // front half: the record at the unit's NEAR boundary
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) brec[qd] = bstate[(size_t)bslot * 256 + qd * 64 + lane];
  const int tx = (int)(where & 0xFFFFu), ty = (int)(where >> 16);
  const int tile = ty * gx + tx;
  const uint32_t start = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  const uint32_t chunks = CHUNKS ? (uint32_t)CHUNKS : meta[2], seg_len = chunks * GS_SEG;   // instances per unit
  const uint32_t boff = seg * seg_len;  // contributor index (0-based) of this unit's first instance
  float4 (*__restrict__ recl)[3] = s_rec[wave];

  // ---- the lane's four pixels
  const int px0 = tx * GS_TILE + (lane & 7), py0 = ty * GS_TILE + (lane >> 3);
  const float fx0 = (float)px0, fy0 = (float)py0;
  const size_t plane = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  float Tr[4], behind[4], g0[4], g1[4], g2[4];
  float oc0[4], oc1[4], oc2[4];   // the frame's colour at the pixel
  int lastq[4];        // the pixel's last contributor (1-based position in the tile list; 0: none)
  uint32_t wmaxq[4];   // wave-wide max of `last` per quadrant
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int px = px0 + (qd & 1) * 8, py = py0 + (qd >> 1) * 8;
    const size_t pix = (size_t)min(py, H - 1) * W + min(px, W - 1);
    Tr[qd] = final_T[pix]; lastq[qd] = (int)n_contrib[pix];
    g0[qd] = dL_dpix[pix]; g1[qd] = dL_dpix[plane + pix]; g2[qd] = dL_dpix[2 * plane + pix];
    oc0[qd] = out_color[pix]; oc1[qd] = out_color[plane + pix]; oc2[qd] = out_color[2 * plane + pix];
  }
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    GS_PIN4(brec[qd].x, brec[qd].y, brec[qd].z, brec[qd].w);
    GS_PIN4(oc0[qd], oc1[qd], oc2[qd], Tr[qd]);
    const int px = px0 + (qd & 1) * 8, py = py0 + (qd >> 1) * 8;
    if (!(px < W && py < H)) { Tr[qd] = 0.f; lastq[qd] = 0; g0[qd] = g1[qd] = g2[qd] = 0.f; }   // outside the image
  }
#undef GS_PIN4
  const uint32_t hlo = boff + ((HALVES && !fwd_dir) ? (uint32_t)GS_SEG / 2u : 0u);   // first instance this wave replays
  if (end <= start + hlo) return;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    behind[qd] = Tr[qd] * (bg0 * g0[qd] + bg1 * g1[qd] + bg2 * g2[qd]);  // dL/dC . (everything behind, background included)
    wmaxq[qd] = qmax[(size_t)tile * 4 + qd];   // the forward's wave maximum of `last` over the quadrant: a scalar load
  }
  // the tile only needs instances [0, max over pixels of last)
  const uint32_t tile_max = min(max(max(wmaxq[0], wmaxq[1]), max(wmaxq[2], wmaxq[3])), end - start);
  if (tile_max <= hlo) return;  // every pixel's last contributor lies in front of this segment
  if (fwd_dir) {
    // front to back from the near boundary: T in front of the unit, and dL/dC . (C_out - C accumulated in front of it) —
    // everything from this unit's first instance to the background
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const bool first = seg == 0u;
      const float c0 = first ? 0.f : brec[qd].y, c1 = first ? 0.f : brec[qd].z, c2 = first ? 0.f : brec[qd].w;
      Tr[qd] = first ? 1.f : brec[qd].x;
      behind[qd] = (oc0[qd] - c0) * g0[qd] + (oc1[qd] - c1) * g1[qd] + (oc2[qd] - c2) * g2[qd];
    }
  } else if (boff + seg_len < tile_max) {
    // not the deepest active unit of the tile: resume from the forward's record at this unit's far boundary
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      Tr[qd] = brec[qd].x;
      behind[qd] = (oc0[qd] - brec[qd].y) * g0[qd] + (oc1[qd] - brec[qd].z) * g1[qd] + (oc2[qd] - brec[qd].w) * g2[qd];
    }
  }

  typedef float v4f __attribute__((vector_size(16)));
  [[maybe_unused]] float sel[9];   // BW_REDUCE_MFMA: the B operands "column c"
  if constexpr (BW_REDUCE_MFMA) {
#pragma unroll
    for (int c = 0; c < 9; ++c) sel[c] = (lane & 15) == c ? 1.f : 0.f;
  }
  const bool bit0 = (lane & 1) != 0, bit1 = (lane & 2) != 0;
  const bool out_lane = (lane & 14) == 0 || lane == 2;                 // the nine lanes that hold a finished sum
  const int out_comp = lane == 2 ? 8 : (lane >> 4) + 4 * (lane & 1);   // ... and which of the nine it is
  unsigned long long mq[4];  // per quadrant: which of the staged chunk's records reach it
  uint32_t cb = 0;           // contributor index (0-based) of the staged chunk's first instance
  [[maybe_unused]] unsigned long long c_steps = 0, c_quads = 0, c_quads_valid = 0, c_lanes = 0, c_reduced = 0;

  // One (Gaussian, tile) step of the back-to-front replay: the pixels of every quadrant the Gaussian reaches, then ONE
  // reduction + one row of nine atomics.
  auto replay_one = [&](const float2& a0, const float4& a1, const float4& a2, const int i2) {
    const uint32_t id = __float_as_uint(a2.w);
    const float dx0 = a0.x - fx0, dy0 = a0.y - fy0;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f, m6 = 0.f, m7 = 0.f, m8 = 0.f;
    bool any_valid = false;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      if ((mq[qd] >> i2) & 1ull) {   // wave-uniform: scalar bit test
        const float dx = (qd & 1) ? dx0 - 8.f : dx0, dy = (qd >> 1) ? dy0 - 8.f : dy0;
        const float power2 = gs_power2(dx, dy, a1.x, a1.y, a1.z);  // log2 of the falloff, the forward's bits
        // opacity * G, unclamped: the reference clamps alpha to 0.99 but lets dL/dalpha through to G unchanged, so
        // G * dL/dG = (opacity G) dL/dalpha needs the unclamped product.  power2 > 0 can make it inf; such lanes are
        // invalid and every use below selects, never multiplies, them away.
        const float au = a1.w * gs_exp2(power2);
        // (contributor = cb + i2 + 1 <= last: the instance lies at or in front of the pixel's last contributor)
        const bool valid = (int)cb + i2 < lastq[qd] && power2 <= 0.0f && au >= ALPHA_MIN;
        if constexpr (COUNT) { c_quads += 1; c_lanes += __popcll(__ballot(valid)); }
        if (__any(valid)) {
          any_valid = true;
          if constexpr (COUNT) c_quads_valid += 1;
          // A lane that must skip this Gaussian treats it as fully transparent (alpha 0): T and the colour behind then
          // evolve exactly as if it had been skipped, so the replay state needs no per-field selects.
          const float av = valid ? au : 0.f;
          const float al = __builtin_amdgcn_fmed3f(av, 0.0f, 0.99f);     // min(0.99, av) for av >= 0: one v_med3_f32
          const float inv_one_m = gs_rcp(1.f - al);                         // v_rcp_f32 (1 ulp) instead of two IEEE divisions
          // dC/dalpha_k = c_k T_k - (sum_{j behind k} c_j alpha_j T_j + T_final bg) / (1 - alpha_k).  Contracted with
          // dL/dC first, the "colour behind" term is ONE running scalar (behind) instead of the reference's three-channel
          // accum_rec / last_color / last_alpha recursion (same quantity: accum_rec_k = sum_{j>k} c_j alpha_j T_j / T_{k+1}).
          const float cg = a2.x * g0[qd] + a2.y * g1[qd] + a2.z * g2[qd];
          float dL_dalpha, dchannel;
          if (HALVES && fwd_dir) {
            // front to back (wave-uniform branch): Tr is the transmittance in FRONT of this Gaussian already, `behind` still
            // contains this Gaussian's own contribution — take it out, then step T forward with the forward's own product
            dchannel = al * Tr[qd];
            const float bk = behind[qd] - cg * dchannel;
            dL_dalpha = Tr[qd] * cg - bk * inv_one_m;
            behind[qd] = bk;
            Tr[qd] = Tr[qd] * (1.f - al);
          } else {
            Tr[qd] = Tr[qd] * inv_one_m;                                  // transmittance in front of this Gaussian
            dL_dalpha = Tr[qd] * cg - behind[qd] * inv_one_m;
            dchannel = al * Tr[qd];
            behind[qd] += cg * dchannel;
          }
          // Moments of w = G * dL/dG over the Gaussian's pixels: every screen-space gradient of this Gaussian is a fixed
          // linear combination of them (coefficients = its own conic / opacity), applied once per Gaussian in
          // k_preprocess_bwd instead of once per pixel here:
          //   dL/dconic = (-1/2 sum w dx^2, -sum w dx dy, -1/2 sum w dy^2),  dL/dopacity = sum w / opacity,
          //   dL/dmean2D = -(a sum w dx + b sum w dy, c sum w dy + b sum w dx) * (W/2, H/2)
          const float w = av * dL_dalpha;
          const float wdx = w * dx, wdy = w * dy;
          m0 += wdx; m1 += wdy;
          m2 = fmaf(wdx, dx, m2); m3 = fmaf(wdx, dy, m3); m4 = fmaf(wdy, dy, m4);
          m5 += w;
          m6 = fmaf(dchannel, g0[qd], m6); m7 = fmaf(dchannel, g1[qd], m7); m8 = fmaf(dchannel, g2[qd], m8);  // dL/drgb
        }
      }
    }
    if constexpr (COUNT) { c_steps += 1; c_reduced += any_valid ? 1 : 0; }
#ifdef GS_PROBE
    pr_steps += 1;
#endif
    if constexpr (BW_REDUCE_LDS) {
      if (any_valid) {
        float* __restrict__ red = s_red[wave];
        // (LDS operations of one wave execute in program order: the reads below see this step's writes, and the next step's
        // writes come after these reads — no barrier; the fences keep the compiler from reordering across lanes' accesses)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        red[0 * RED_PITCH + lane] = m0; red[1 * RED_PITCH + lane] = m1; red[2 * RED_PITCH + lane] = m2;
        red[3 * RED_PITCH + lane] = m3; red[4 * RED_PITCH + lane] = m4; red[5 * RED_PITCH + lane] = m5;
        red[6 * RED_PITCH + lane] = m6; red[7 * RED_PITCH + lane] = m7; red[8 * RED_PITCH + lane] = m8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float part = 0.f;
        if (lane < 36) {
          const float4* q = reinterpret_cast<const float4*>(red + (lane >> 2) * RED_PITCH + (lane & 3) * 16);
          const float4 a = q[0], b = q[1], c = q[2], d = q[3];
          part = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
        }
        part += gs_dpp<0xB1>(part);   // quad_perm [1,0,3,2]
        part += gs_dpp<0x4E>(part);   // quad_perm [2,3,0,1]: every lane of the quad holds the moment's sum over the wave
        if constexpr (DET) {
          if ((lane & 3) == 0 && lane < 36 && id < capacity) det_rows[(size_t)id * 12 + (lane >> 2)] = part;   // id: the instance's row
        } else {
          if ((lane & 3) == 0 && lane < 36) atomicAdd(reinterpret_cast<float*>(grads + id) + (lane >> 2), part);
        }
      }
    } else if constexpr (BW_REDUCE_MFMA) {
      if (any_valid) {
        // EXPERIMENT: the cross-row half of the reduction on the matrix pipe.  v_mfma_f32_16x16x4_f32 computes
        // D[i][j] += sum_k A[i][k] B[k][j] with A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[4 R + t][j] in register t of
        // lane 16 R + j: with A = a moment and B = "column c" (1 in lanes with lane % 16 == c) the four rows of 16 lanes are summed
        // and moment c lands in column c; nine accumulating MFMAs put the nine moments side by side.  Exact fp32.
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m0, sel[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1, sel[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m2, sel[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m3, sel[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m4, sel[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m5, sel[5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m6, sel[6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m7, sel[7], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m8, sel[8], acc, 0, 0, 0);
        // lane (R, c): register t holds the sum over the four rows of lanes 4 R + t (mod 16) -> add the registers, then the rows
        const float mine = gs_sum_rows((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (lane < 9) atomicAdd(reinterpret_cast<float*>(grads + id) + lane, mine);
      }
    } else if (any_valid) {
      // Nine values x 64 lanes -> nine sums, transposed so that every step halves the number of live values:
      // rows first (v_permlane16/32_swap pair steps, two ops per pair), then lane bits 0 and 1 inside the row (DPP
      // quad_perm pair steps), then the four quads of the row (row_ror:4, row_ror:8) on the single survivor.
      const float u0 = gs_pair_reduce_rows16(m0, m1), u1 = gs_pair_reduce_rows16(m2, m3);
      const float u2 = gs_pair_reduce_rows16(m4, m5), u3 = gs_pair_reduce_rows16(m6, m7);
      const float m8o = gs_opaque(m8);
      const float u4 = gs_pair_reduce_rows16(m8o, m8o);
      const float v0 = gs_pair_reduce_rows32(u0, u1);  // row r holds component r     (m0..m3)
      const float v1 = gs_pair_reduce_rows32(u2, u3);  // row r holds component 4 + r (m4..m7)
      const float v2 = gs_pair_reduce_rows32(u4, u4);  // every row holds component 8
      const float y0 = gs_pair_reduce<0xB1>(bit0, v0, v1);
      const float y1 = v2 + gs_dpp<0xB1>(v2);
      float mine = gs_pair_reduce<0x4E>(bit1, y0, y1);
      mine += gs_dpp<0x124>(mine);
      mine += gs_dpp<0x128>(mine);
      // lanes 16r and 16r+1 now hold components r and 4+r, lane 2 holds component 8
      if (out_lane) atomicAdd(reinterpret_cast<float*>(grads + id) + out_comp, mine);
    }
  };

#ifdef GS_PROBE
  pr_t1 = GS_PROBE_CLOCK();
#endif
  // ---- the unit's chunks of GS_SEG instances, deepest first
#pragma unroll 1
  for (int c = (int)chunks - 1; c >= 0; --c) {
    cb = boff + (uint32_t)c * GS_SEG;
    if (cb >= tile_max) continue;
    // stage the chunk's records (lane i <- instance cb + i); only this wave reads them back
    const int cnt = (int)min((uint32_t)GS_SEG, tile_max - cb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the previous chunk's broadcast reads are done
    if (lane < cnt) {
      const uint32_t id = list[start + cb + lane];
      const GsRec* r = recs + id;
      const float4 col = r->q2;  // (r, g, b, depth): depth is not used here, its slot carries the Gaussian's index
      recl[lane][0] = r->q0; recl[lane][1] = r->q1;
      recl[lane][2] = make_float4(col.x, col.y, col.z, __uint_as_float(DET ? det_rowidx[start + cb + lane] : id));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // cull: the forward has tested every staged record against the four quadrants (quad_hit) and left the four masks of this
    // chunk; instances behind every pixel of a quadrant are dropped here (scalar mask arithmetic).  A quadrant whose forward
    // wave stopped before this chunk never wrote its mask — and has no pixel that reaches the chunk (wmaxq <= cb): masked off.
    {
      const uint32_t chunk = slot * chunks + (uint32_t)c;
      const unsigned long long* hm = hitmask + (size_t)min(chunk, max_chunks - 1u) * 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const uint32_t reach = wmaxq[qd] > cb ? wmaxq[qd] - cb : 0u;   // instances of the chunk at or in front of the quadrant's last contributor
        const unsigned long long keep = reach >= 64u ? ~0ull : ((1ull << reach) - 1ull);
        mq[qd] = chunk < max_chunks ? (hm[qd] & keep) : 0ull;
      }
    }
    if constexpr (HALVES) {
      const unsigned long long mine = fwd_dir ? 0x00000000ffffffffull : 0xffffffff00000000ull;   // this wave's half of the unit
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) mq[qd] &= mine;
    }
    unsigned long long many = (mq[0] | mq[1]) | (mq[2] | mq[3]);
    if (!many) continue;
    // the next instance of the walk: the deepest remaining one back to front, the nearest one front to back (scalar)
    auto top = [&](unsigned long long m) { return (HALVES && fwd_dir) ? (int)__builtin_ctzll(m) : 63 - __clzll((long long)m); };
    // back-to-front walk over the union mask, unrolled by two with ping-pong record registers: the next record's LDS reads
    // (wave-uniform addresses: broadcasts) are issued before the current record's math
    auto xy = [&](int i) { const float4& r = recl[i][0]; return make_float2(r.x, r.y); };   // the walk only needs the centre
    int iA = top(many), iB = iA;
    float2 A0 = xy(iA), B0 = A0;
    float4 A1 = recl[iA][1], A2 = recl[iA][2], B1 = A1, B2 = A2;
    for (;;) {
      many &= ~(1ull << iA);
      const bool moreB = many != 0;
      if (moreB) iB = top(many);
      B0 = xy(iB); B1 = recl[iB][1]; B2 = recl[iB][2];
      replay_one(A0, A1, A2, iA);
      if (!moreB) break;
      many &= ~(1ull << iB);
      const bool moreA = many != 0;
      if (moreA) iA = top(many);
      A0 = xy(iA); A1 = recl[iA][1]; A2 = recl[iA][2];
      replay_one(B0, B1, B2, iB);
      if (!moreA) break;
    }
  }
  GS_PROBE_STORE(4096u + (HALVES ? hu : unit), pr_t0, GS_PROBE_CLOCK(), pr_steps, pr_t1 - pr_t0, (unsigned long long)tile, (unsigned long long)seg,
                 (unsigned long long)gs_physical_cu(), (unsigned long long)blockIdx.x);
  if constexpr (COUNT) {
    if (lane == 0 && counters) {
      atomicAdd(counters + 0, c_steps); atomicAdd(counters + 1, c_quads); atomicAdd(counters + 2, c_quads_valid);
      atomicAdd(counters + 3, c_lanes); atomicAdd(counters + 4, c_reduced); atomicAdd(counters + 5, 1ull);
    }
  }
}

// ---- deterministic mode: rows of the instances and their per-Gaussian sums --------------------------------------------------
// one workgroup per tile: the row of list position p (Gaussian g in tile (tx, ty)) = off[g] + the tile's index in g's rectangle
__global__ __launch_bounds__(256) void k_det_rowidx(int gx, uint32_t capacity, const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ list, const uint2* __restrict__ rects,
                                                    const uint32_t* __restrict__ off, uint32_t* __restrict__ rowidx) {
  const int tile = blockIdx.x, tx = tile % gx, ty = tile / gx;
  const uint32_t s = min(tile_start[tile], capacity), e = min(tile_start[tile + 1], capacity);
  for (uint32_t p = s + threadIdx.x; p < e; p += 256) {
    const uint32_t g = list[p];
    const uint2 r = rects[g];
    const int x0 = r.x & 0xffff, y0 = r.x >> 16, x1 = r.y & 0xffff;
    rowidx[p] = off[g] + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
  }
}
// twelve lanes per Gaussian (one per float of its GsGrad record; nine carry sums): the rows of its instances, contiguous from
// off[g], added in rectangle order (y outer, x inner) — a fixed order.  Rows the backward never wrote are the zeros the memset left.
__global__ __launch_bounds__(192) void k_det_gather(int P, uint32_t capacity, const uint32_t* __restrict__ off, const float* __restrict__ rows,
                                                    GsGrad* __restrict__ grads) {
  const int g = blockIdx.x * 16 + threadIdx.x / 12, c = threadIdx.x % 12;
  if (g >= P) return;
  const uint32_t r0 = min(off[g], capacity), r1 = min(off[g + 1], capacity);
  const float* __restrict__ src = rows + (size_t)r0 * 12 + c;
  float acc = 0.f;
  uint32_t n = r1 - r0;
  // eight rows in flight, added in row order (the ORDER is what the mode is about; the loads may run ahead)
  for (; n >= 8u; n -= 8u, src += 96) {
    const float a0 = src[0], a1 = src[12], a2 = src[24], a3 = src[36], a4 = src[48], a5 = src[60], a6 = src[72], a7 = src[84];
    acc = (((((((acc + a0) + a1) + a2) + a3) + a4) + a5) + a6) + a7;
  }
  for (; n > 0u; --n, src += 12) acc += *src;
  reinterpret_cast<float*>(grads + g)[c] = acc;
}

// per-tile max of n_contrib -> R_eff (roofline accounting only)
__global__ __launch_bounds__(256) void k_frame_stats(int T, int gx, int W, int H, const uint32_t* __restrict__ tile_start,
                                                      const uint32_t* __restrict__ n_contrib, int64_t* __restrict__ stats) {
  __shared__ uint32_t s_max[4];
  const int tile = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const Quad q = make_quad(tile, gx, W, H);
  uint32_t v = q.inside ? n_contrib[(size_t)q.py * W + q.px] : 0u;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
  if (lane == 0) s_max[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    atomicAdd(reinterpret_cast<unsigned long long*>(stats + 1), (unsigned long long)m);
    if (tile == 0) stats[0] = (int64_t)tile_start[T];
  }
}

}  // namespace

int gs_launch_scan_rect_areas(hipStream_t, int, const uint2*, uint32_t*, uint32_t*, int32_t*);
// deterministic mode, in front of the composite backward: row offsets (a scan over the rectangles' tile counts) -> the row of
// every list position; the rows cleared.  scratch: DetScratchLayout(P) (uint32 words behind the gradient scratch's gate flags).
int gs_launch_det_prepare(hipStream_t stream, int P, int T, int gx, uint32_t capacity, const uint32_t* tile_start, const uint32_t* list,
                          const uint2* rects, char* det_scratch, uint32_t* rowidx, float* rows) {
  const DetScratchLayout dl(P);
  uint32_t* off = (uint32_t*)(det_scratch + dl.off);
  gs_launch_scan_rect_areas(stream, P, rects, off, (uint32_t*)(det_scratch + dl.block_sums), (int32_t*)(off + P + 1));
  hipLaunchKernelGGL(k_det_rowidx, dim3(T), dim3(256), 0, stream, gx, capacity, tile_start, list, rects, off, rowidx);
  if (hipMemsetAsync(rows, 0, (size_t)capacity * sizeof(GsGrad), stream) != hipSuccess) return -1;
  return 0;
}
int gs_launch_det_gather(hipStream_t stream, int P, uint32_t capacity, const char* det_scratch, const float* rows, GsGrad* grads) {
  const DetScratchLayout dl(P);
  hipLaunchKernelGGL(k_det_gather, dim3((P + 15) / 16), dim3(192), 0, stream, P, capacity, (const uint32_t*)(det_scratch + dl.off), rows, grads);
  return 0;
}

int gs_launch_frame_stats(hipStream_t stream, int T, int gx, int W, int H, const uint32_t* tile_start, const uint32_t* n_contrib,
                          int64_t* stats) {
  hipLaunchKernelGGL(k_frame_stats, dim3(T), dim3(256), 0, stream, T, gx, W, H, tile_start, n_contrib, stats);
  return 0;
}

int gs_launch_composite_fwd(hipStream_t stream, int T, int gx, int W, int H, uint32_t capacity, const uint32_t* tile_start,
                            const uint32_t* list, const GsRec* recs, const float* bg, float* out_color, float* final_T,
                            uint32_t* n_contrib, const uint32_t* order, const uint32_t* seg_first, const uint32_t* part_first,
                            uint4* unit_tile, float4* bstate, uint32_t max_units, const uint32_t* meta, unsigned long long* hitmask,
                            uint32_t max_chunks, uint32_t* qmax, unsigned long long* counters, bool train) {
  // equal-weight workgroups of 4 (2) tiles while the whole frame is one resident round of at most one (two) workgroups per CU
  // (the count belongs to the device the launch goes to — the process's current one — not to whichever device was current
  // the first time: a process that drives two different GPUs gets each one's own; a device attribute read is host-only and cheap,
  // the table just saves it)
  static int cus_of[64] = {};   // racing writers store the same value
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int cus = cus_of[dev];
  if (cus == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus_of[dev] = cus = n;
  }
  const int per_wg = GS_FWD_TILES ? GS_FWD_TILES : (T <= 4 * cus ? 4 : (T <= 4 * cus * 2 ? 2 : 1));
  // GS_FWD_LDS_PAD (A/B builds): unused dynamic LDS per workgroup, which bounds how many workgroups a CU holds at once — fewer
  // resident waves than work items, so that the dispatcher hands the lightest tiles to whichever CU frees a slot first
#ifndef GS_FWD_LDS_PAD
#define GS_FWD_LDS_PAD 0
#endif
#define GS_FWD(N, CNT, TRN)                                                                                                         \
  hipLaunchKernelGGL((k_composite_fwd<N, CNT, TRN>), dim3((T + N - 1) / N), dim3(256 * N), GS_FWD_LDS_PAD, stream, T, gx, W, H, capacity, tile_start, list, recs, bg, \
                     out_color, final_T, n_contrib, order, seg_first, part_first, unit_tile, bstate, max_units, meta, hitmask, max_chunks, qmax, counters)
  if (!train) { if (per_wg == 4) GS_FWD(4, false, false); else if (per_wg == 2) GS_FWD(2, false, false); else GS_FWD(1, false, false); }
  else if (counters) { if (per_wg == 4) GS_FWD(4, true, true); else if (per_wg == 2) GS_FWD(2, true, true); else GS_FWD(1, true, true); }
  else { if (per_wg == 4) GS_FWD(4, false, true); else if (per_wg == 2) GS_FWD(2, false, true); else GS_FWD(1, false, true); }
#undef GS_FWD
  return 0;
}

// one wave per backward unit (BW_UNITS per workgroup); the grid covers every unit the buffers can hold, waves past the
// frame's count exit
int gs_launch_composite_bwd(hipStream_t stream, int gx, int W, int H, uint32_t capacity, const uint32_t* tile_start,
                            const uint32_t* list, const GsRec* recs, const float* bg, const float* final_T,
                            const uint32_t* n_contrib, const float* dL_dpix, GsGrad* grads, const float* out_color,
                            const uint4* unit_tile, const float4* bstate, const uint32_t* meta,
                            uint32_t max_units, bool may_loop, const unsigned long long* hitmask, uint32_t max_chunks,
                            const uint32_t* qmax, unsigned long long* counters, const uint32_t* det_rowidx, float* det_rows) {
  // whole blocks of launch positions (see the index transposition in the kernel); with GS_BW_HALF the one-chunk instantiation
  // takes two positions per unit
  const uint32_t per_unit = (GS_BW_HALF != 0 && !may_loop) ? 2u : 1u;
  const uint32_t blk = 8u * BW_XCD_RUN * per_unit;
  const dim3 grid((max_units * per_unit + blk - 1u) / blk * blk);
  // may_loop == false: a frame that fits this capacity has one-chunk units (count <= capacity), so the lean instantiation is safe
#define GS_BWD(CH, CNT, DT)                                                                                                           \
  hipLaunchKernelGGL((k_composite_bwd<CH, CNT, DT>), grid, dim3(64 * BW_UNITS), 0, stream, gx, W, H, capacity, tile_start, list, recs, bg, final_T,  \
                     n_contrib, dL_dpix, grads, out_color, unit_tile, bstate, meta, max_units, counters, hitmask, max_chunks, qmax, det_rowidx, det_rows)
  if (det_rows) { if (!may_loop) GS_BWD(1, false, true); else GS_BWD(0, false, true); }
  else if (counters) { if (!may_loop) GS_BWD(1, true, false); else GS_BWD(0, true, false); }
  else { if (!may_loop) GS_BWD(1, false, false); else GS_BWD(0, false, false); }
#undef GS_BWD
  return 0;
}
