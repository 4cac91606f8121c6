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

namespace {

constexpr int BATCH = 512;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

struct Quad {
  int px, py;          // this lane's pixel
  gs_v2f f;            // as float (x, y)
  float x0, x1, y0, y1;  // the wave's 8x8 pixel box (inclusive, float)
  bool inside;
};

__device__ __forceinline__ Quad make_quad_xy(int tx, int ty, int W, int H) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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

// ------------------------------------------------------------------------------------------------
// K6 forward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void composite_fwd_tile(int tile, int gx, int W, int H, uint32_t capacity,
                                                   const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ list,
                                                   const GsRec* __restrict__ recs, const float* __restrict__ bg,
                                                   float* __restrict__ out_color, float* __restrict__ final_T,
                                                   uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ seg_first,
                                                   uint2* __restrict__ unit_tile, float4* __restrict__ bstate, uint32_t max_units) {
  __shared__ float4 s_q0[BATCH];
  __shared__ float4 s_q1[BATCH];
  __shared__ float4 s_q2[BATCH];
  __shared__ int s_done[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Quad q = make_quad(tile, gx, W, H);
  const uint32_t start = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  // Backward units of this tile (segments of GS_SEG instances, see common.h): publish them, and leave every pixel's
  // (transmittance after the last blended Gaussian, accumulated colour) at each segment boundary for the backward.
  const uint32_t seg0 = seg_first[tile], nseg = seg_first[tile + 1] - seg0;
  for (uint32_t sg = tid; sg < nseg; sg += 256)
    if (seg0 + sg < max_units)  // (tile x | tile y << 16, segment): the backward needs no division to place itself
      unit_tile[seg0 + sg] = make_uint2((uint32_t)(tile % gx) | ((uint32_t)(tile / gx) << 16), sg);
  uint32_t next_boundary = 0;  // boundaries [0, next_boundary) of this tile have been stored by this wave

  // Tr: live transmittance, forced to 0 once the pixel is finished (T < 1e-4 reached, or outside the image);
  // Tfin: transmittance after the last blended Gaussian (the value the reference stores as final_T)
  float Tr = q.inside ? 1.0f : 0.0f, Tfin = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;

  for (uint32_t base = start; base < end; base += BATCH) {
    const int wave_done = __all(Tr == 0.0f);
    if (lane == 0) s_done[wave] = wave_done;
    __syncthreads();  // also fences the previous batch's LDS reads against the stores below
    if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
#pragma unroll
    for (int sl = tid; sl < BATCH; sl += 256) {
      const uint32_t j = base + sl;
      if (j < end) {
        const GsRec* r = recs + list[j];
        s_q0[sl] = r->q0; s_q1[sl] = r->q1; s_q2[sl] = r->q2;
      }
    }
    __syncthreads();
    if (wave_done) continue;
    const int cnt = (int)min((uint32_t)BATCH, end - base);
    for (int k = 0; k < cnt; k += 64) {
      const int i = k + lane;
      bool hit = false;
      if (i < cnt) hit = quad_hit(s_q0[i], s_q1[i], q);
      unsigned long long mask = __ballot(hit);
      if (mask) {
        // Software-pipelined walk over the hit mask, unrolled by two with ping-pong record registers: the next
        // record's LDS reads are issued before the current record's math and no register copies are needed to
        // rotate the prefetch.  The math is predicated (no exec-mask branches).
        auto blend_one = [&](const float4& a0, const float4& a1, const float4& a2, int i2) {
          const gs_v2f d = gs_v2f{a0.x, a0.y} - q.f;
          const gs_v2f sq = (d * gs_v2f{a1.x, a1.y}) * d;                     // packed: (A dx^2, C dy^2)
          const float power2 = fmaf(a1.z * d[0], d[1], sq[0] + sq[1]);        // log2 of the falloff
          const float alpha = fminf(0.99f, a1.w * __builtin_amdgcn_exp2f(power2));
          // a skipped Gaussian is a transparent one; a finished pixel carries Tr == 0, so `stop` (and nothing else)
          // also covers "already done" and no separate flag is tested here
          const float al = (power2 <= 0.0f && alpha >= ALPHA_MIN) ? alpha : 0.0f;
          const float w0 = al * Tr;
          const float test_T = Tr - w0;  // T (1 - alpha)
          const bool stop = test_T < T_MIN;
          const float w = stop ? 0.0f : w0;
          C0 += a2.x * w; C1 += a2.y * w; C2 += a2.z * w;
          last = (w > 0.0f) ? (base - start) + (uint32_t)i2 + 1u : last;  // blended: alpha > 0 and not the stopping one
          Tfin = stop ? Tfin : test_T;
          Tr = stop ? 0.0f : test_T;
        };
        int iA = k + __ffsll(mask) - 1, iB = iA;
        float4 A0 = s_q0[iA], A1 = s_q1[iA], A2 = s_q2[iA], B0 = A0, B1 = A1, B2 = A2;
        for (;;) {
          mask &= mask - 1;
          const bool moreB = mask != 0;
          if (moreB) iB = k + __ffsll(mask) - 1;
          B0 = s_q0[iB]; B1 = s_q1[iB]; B2 = s_q2[iB];
          blend_one(A0, A1, A2, iA);
          if (!moreB) break;
          mask &= mask - 1;
          const bool moreA = mask != 0;
          if (moreA) iA = k + __ffsll(mask) - 1;
          A0 = s_q0[iA]; A1 = s_q1[iA]; A2 = s_q2[iA];
          blend_one(B0, B1, B2, iB);
          if (!moreA) break;
        }
      }
      const uint32_t pos = (base - start) + (uint32_t)k + 64u;  // instances of the tile blended so far
      if (pos % GS_SEG == 0u) {
        next_boundary = pos / GS_SEG;
        if (seg0 + next_boundary - 1u < max_units) bstate[(size_t)(seg0 + next_boundary - 1u) * 256 + tid] = make_float4(Tfin, C0, C1, C2);
      }
      if (__all(Tr == 0.0f)) break;
    }
  }
  // boundaries this wave never reached (all its pixels were finished, or the tile ended): the state no longer changes
  for (uint32_t sg = next_boundary; sg + 1u < nseg; ++sg)
    if (seg0 + sg < max_units) bstate[(size_t)(seg0 + sg) * 256 + tid] = make_float4(Tfin, C0, C1, C2);
  if (q.inside) {
    const size_t pix = (size_t)q.py * W + q.px, plane = (size_t)W * H;
    final_T[pix] = Tfin;
    n_contrib[pix] = last;
    out_color[pix] = C0 + Tfin * bg[0];
    out_color[plane + pix] = C1 + Tfin * bg[1];
    out_color[2 * plane + pix] = C2 + Tfin * bg[2];
  }
}

__global__ __launch_bounds__(256) void k_composite_fwd(int T, int gx, int W, int H, uint32_t capacity,
                                                        const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ list,
                                                        const GsRec* __restrict__ recs, const float* __restrict__ bg,
                                                        float* __restrict__ out_color, float* __restrict__ final_T,
                                                        uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order,
                                                        GsSched* sched, int NB, const uint32_t* __restrict__ seg_first,
                                                        uint2* __restrict__ unit_tile, float4* __restrict__ bstate, uint32_t max_units) {
  GS_PERSISTENT_TILE_LOOP(sched, NB, T, order,
                          composite_fwd_tile(tile, gx, W, H, capacity, tile_start, list, recs, bg, out_color, final_T, n_contrib,
                                             seg_first, unit_tile, bstate, max_units))
}

// ------------------------------------------------------------------------------------------------
// K7 backward: replay each tile back to front.  Per (pixel, Gaussian) the nine screen-space
// gradient terms are reduced across the wave's 64 pixels in registers (butterfly shuffles) and
// leave the wave as ONE set of float atomics per Gaussian per wave (the reference operator issues
// them per pixel).  Gaussians whose box misses the wave's quadrant, or that lie behind every
// pixel's last contributor, are skipped wave-wide.
// ------------------------------------------------------------------------------------------------
// One workgroup = one UNIT: segment `seg` (GS_SEG instances) of one tile's list.  The state a back-to-front replay would
// carry into the segment comes from the forward's boundary record instead: T is the forward's own product, and the colour
// behind is dL/dC . (final colour - colour accumulated in front of the boundary).
__global__ __launch_bounds__(256) void k_composite_bwd(int gx, int W, int H, uint32_t capacity,
                                                        const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ list,
                                                        const GsRec* __restrict__ recs, const float* __restrict__ bg,
                                                        const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix, GsGrad* __restrict__ grads,
                                                        const float* __restrict__ out_color, const uint32_t* __restrict__ seg_first,
                                                        const uint2* __restrict__ unit_tile, const float4* __restrict__ bstate,
                                                        const uint32_t* __restrict__ meta, uint32_t max_units) {
  constexpr int BATCH = GS_SEG;
  __shared__ float4 s_q0[BATCH];
  __shared__ float4 s_q1[BATCH];
  __shared__ float4 s_q2[BATCH];
  __shared__ uint32_t s_max[4];
  const uint32_t unit = blockIdx.x;
  if (unit >= min(meta[1], max_units)) return;
  const uint2 entry = unit_tile[unit];
  const uint32_t where = __builtin_amdgcn_readfirstlane(entry.x), seg = __builtin_amdgcn_readfirstlane(entry.y);  // uniform: scalar
  const int tx = (int)(where & 0xFFFFu), ty = (int)(where >> 16);
  const int tile = ty * gx + tx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Quad q = make_quad_xy(tx, ty, W, H);
  const uint32_t start = min(tile_start[tile], capacity), end = min(tile_start[tile + 1], capacity);
  const uint32_t boff = seg * GS_SEG;  // contributor index (0-based) of this unit's first instance
  if (end <= start + boff) return;

  const size_t pix = (size_t)q.py * W + q.px, plane = (size_t)W * H;
  const float T_final = q.inside ? final_T[pix] : 0.f;
  const uint32_t last = q.inside ? n_contrib[pix] : 0u;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (q.inside) { g0 = dL_dpix[pix]; g1 = dL_dpix[plane + pix]; g2 = dL_dpix[2 * plane + pix]; }
  const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
  const gs_v2f g01 = {g0, g1};

  // the tile only needs instances [0, max over pixels of last)
  const uint32_t wmax = gs_wave_max_u32(last);
  if (lane == 0) s_max[wave] = wmax;
  __syncthreads();
  const uint32_t tile_max = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), end - start);
  if (tile_max <= boff) return;  // every pixel's last contributor lies in front of this segment

  const bool bit0 = (lane & 1) != 0, bit1 = (lane & 2) != 0;
  const bool out_lane = (lane & 14) == 0 || lane == 2;                 // the nine lanes that hold a finished sum
  const int out_comp = lane == 2 ? 8 : (lane >> 4) + 4 * (lane & 1);   // ... and which of the nine it is
  float Tr = T_final;
  float behind = T_final * bg_dot;  // dL/dC . (everything composited behind the current Gaussian, background included)
  if (boff + GS_SEG < tile_max) {
    // not the deepest active segment: resume from the forward's record at this segment's far boundary
    const uint32_t slot = seg_first[tile] + seg;
    if (slot < max_units) {
      const float4 b = bstate[(size_t)slot * 256 + tid];
      Tr = b.x;
      float c0 = 0.f, c1 = 0.f, c2 = 0.f;
      if (q.inside) { c0 = out_color[pix]; c1 = out_color[plane + pix]; c2 = out_color[2 * plane + pix]; }
      behind = (c0 - b.y) * g0 + (c1 - b.z) * g1 + (c2 - b.w) * g2;
    }
  }

  {
#pragma unroll
    for (int sl = tid; sl < BATCH; sl += 256) {
      if (boff + sl < tile_max) {
        const uint32_t id = list[start + boff + sl];
        const GsRec* r = recs + id;
        const float4 c = r->q2;  // (r, g, b, depth): depth is not used here, its slot carries the Gaussian's index
        s_q0[sl] = r->q0; s_q1[sl] = r->q1; s_q2[sl] = make_float4(c.x, c.y, c.z, __uint_as_float(id));
      }
    }
    __syncthreads();
    if (boff >= wmax) return;  // nothing in this segment is in front of any of this wave's pixels' last contributor
    const int cnt = (int)min((uint32_t)BATCH, tile_max - boff);
    for (int k = ((cnt - 1) >> 6) << 6; k >= 0; k -= 64) {
      if (boff + (uint32_t)k >= wmax) continue;
      const int i = k + lane;
      bool hit = false;
      if (i < cnt && boff + (uint32_t)i < wmax) hit = quad_hit(s_q0[i], s_q1[i], q);
      unsigned long long mask = __ballot(hit);
      if (mask) {
        // back-to-front walk over the hit mask, unrolled by two with ping-pong record registers (see the forward kernel)
        auto replay_one = [&](const float4& a0, const float4& a1, const float4& a2, const int i2) {
          const uint32_t id = __float_as_uint(a2.w);
          const uint32_t contributor = boff + (uint32_t)i2 + 1u;  // 1-based position in the tile list
          const gs_v2f d = gs_v2f{a0.x, a0.y} - q.f;
          const gs_v2f sq = (d * gs_v2f{a1.x, a1.y}) * d;                     // packed: (A dx^2, C dy^2)
          const float power2 = fmaf(a1.z * d[0], d[1], sq[0] + sq[1]);        // log2 of the falloff
          // opacity * G, unclamped: the reference clamps alpha to 0.99 but lets dL/dalpha through to G unchanged, so
          // G * dL/dG = (opacity G) dL/dalpha needs the unclamped product.  power2 > 0 can make it inf; such lanes are
          // invalid and every use below selects, never multiplies, them away.
          const float au = a1.w * __builtin_amdgcn_exp2f(power2);
          const bool valid = contributor <= last && power2 <= 0.0f && au >= ALPHA_MIN;
          if (__any(valid)) {
            // A lane that must skip this Gaussian treats it as fully transparent (alpha 0): T and the colour behind then
            // evolve exactly as if it had been skipped, so the replay state needs no per-field selects.
            const float av = valid ? au : 0.f;
            const float al = fminf(0.99f, av);
            const float inv_one_m = __builtin_amdgcn_rcpf(1.f - al);  // v_rcp_f32 (1 ulp) instead of two IEEE divisions
            Tr = Tr * inv_one_m;                                       // transmittance in front of this Gaussian
            // dC/dalpha_k = c_k T_k - (sum_{j behind k} c_j alpha_j T_j + T_final bg) / (1 - alpha_k).  Contracted with
            // dL/dC first, the "colour behind" term is ONE running scalar (behind) instead of the reference's three-channel
            // accum_rec / last_color / last_alpha recursion (same quantity: accum_rec_k = sum_{j>k} c_j alpha_j T_j / T_{k+1}).
            const float cg = a2.x * g0 + a2.y * g1 + a2.z * g2;
            const float dL_dalpha = Tr * cg - behind * inv_one_m;
            const float dchannel = al * Tr;
            behind += cg * dchannel;
            // Moments of w = G * dL/dG over the wave's pixels: every screen-space gradient of this Gaussian is a fixed
            // linear combination of them (coefficients = its own conic / opacity), applied once per Gaussian in
            // k_preprocess_bwd instead of once per pixel here:
            //   dL/dconic = (-1/2 sum w dx^2, -sum w dx dy, -1/2 sum w dy^2),  dL/dopacity = sum w / opacity,
            //   dL/dmean2D = -(a sum w dx + b sum w dy, c sum w dy + b sum w dx) * (W/2, H/2)
            const float w = av * dL_dalpha;
            const gs_v2f t01 = gs_v2f{w, w} * d;               // sum w dx, sum w dy
            const gs_v2f t24 = t01 * d;                        // sum w dx^2, sum w dy^2
            const float t0 = t01[0], t1 = t01[1], t2 = t24[0], t4 = t24[1];
            const float t3 = t0 * d[1];                        // sum w dx dy
            const float t5 = w;                                // sum w
            const gs_v2f t67 = gs_v2f{dchannel, dchannel} * g01;
            const float t6 = t67[0], t7 = t67[1], t8 = gs_opaque(dchannel * g2);  // dL/drgb
            // Nine values x 64 lanes -> nine sums, transposed so that every step halves the number of live values:
            // rows first (v_permlane16/32_swap pair steps, two ops per pair), then lane bits 0 and 1 inside the row (DPP
            // quad_perm pair steps), then the four quads of the row (row_ror:4, row_ror:8) on the single survivor.
            const float u0 = gs_pair_reduce_rows16(t0, t1), u1 = gs_pair_reduce_rows16(t2, t3);
            const float u2 = gs_pair_reduce_rows16(t4, t5), u3 = gs_pair_reduce_rows16(t6, t7);
            const float u4 = gs_pair_reduce_rows16(t8, t8);
            const float v0 = gs_pair_reduce_rows32(u0, u1);  // row r holds component r     (t0..t3)
            const float v1 = gs_pair_reduce_rows32(u2, u3);  // row r holds component 4 + r (t4..t7)
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
        int iA = k + 63 - __clzll((long long)mask), iB = iA;
        float4 A0 = s_q0[iA], A1 = s_q1[iA], A2 = s_q2[iA], B0 = A0, B1 = A1, B2 = A2;
        for (;;) {
          mask &= ~(1ull << (iA - k));
          const bool moreB = mask != 0;
          if (moreB) iB = k + 63 - __clzll((long long)mask);
          B0 = s_q0[iB]; B1 = s_q1[iB]; B2 = s_q2[iB];
          replay_one(A0, A1, A2, iA);
          if (!moreB) break;
          mask &= ~(1ull << (iB - k));
          const bool moreA = mask != 0;
          if (moreA) iA = k + 63 - __clzll((long long)mask);
          A0 = s_q0[iA]; A1 = s_q1[iA]; A2 = s_q2[iA];
          replay_one(B0, B1, B2, iB);
          if (!moreA) break;
        }
      }
    }
  }
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

int gs_launch_frame_stats(hipStream_t stream, int T, int gx, int W, int H, const uint32_t* tile_start, const uint32_t* n_contrib,
                          int64_t* stats) {
  hipLaunchKernelGGL(k_frame_stats, dim3(T), dim3(256), 0, stream, T, gx, W, H, tile_start, n_contrib, stats);
  return 0;
}

int gs_launch_composite_fwd(hipStream_t stream, int T, int gx, int W, int H, uint32_t capacity, const uint32_t* tile_start,
                            const uint32_t* list, const GsRec* recs, const float* bg, float* out_color, float* final_T,
                            uint32_t* n_contrib, const uint32_t* order, GsSched* sched, const uint32_t* seg_first, uint2* unit_tile,
                            float4* bstate, uint32_t max_units) {
  const int NB = gs_num_cus();
  hipLaunchKernelGGL(k_composite_fwd, dim3(gs_grid_persistent(T, NB)), dim3(256), 0, stream, T, gx, W, H, capacity, tile_start, list,
                     recs, bg, out_color, final_T, n_contrib, order, sched + GS_SCHED_FWD, NB, seg_first, unit_tile, bstate, max_units);
  return 0;
}

// one workgroup per backward unit; the grid covers every unit the buffers can hold, workgroups past the frame's count exit
int gs_launch_composite_bwd(hipStream_t stream, int gx, int W, int H, uint32_t capacity, const uint32_t* tile_start,
                            const uint32_t* list, const GsRec* recs, const float* bg, const float* final_T,
                            const uint32_t* n_contrib, const float* dL_dpix, GsGrad* grads, const float* out_color,
                            const uint32_t* seg_first, const uint2* unit_tile, const float4* bstate, const uint32_t* meta,
                            uint32_t max_units) {
  hipLaunchKernelGGL(k_composite_bwd, dim3(max_units), dim3(256), 0, stream, gx, W, H, capacity, tile_start, list, recs, bg, final_T,
                     n_contrib, dL_dpix, grads, out_color, seg_first, unit_tile, bstate, meta, max_units);
  return 0;
}
