// Fused SSIM (+ L1) loss, forward and backward.
// replaces fused_ssim.fused_ssim at reference train.py:173; value/gradient definition pinned by the
// reference's own fallback utils/loss_utils.py:55-85 (11-tap Gaussian, sigma 1.5, zero "same"
// padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements).  L1 is reference utils/loss_utils.py:39-40.
//
// One workgroup = one 32x16 output tile of one (batch, channel) plane.  The 42x26 input halo of both
// images is staged in LDS once; the 11x11 window is applied separably (row pass into LDS, column
// pass in registers) for the five moments x, y, x^2, y^2, xy.  A thread owns the output columns
// (lx, lx + 16) and carries them as one packed-FP32 pair through both passes (v_pk_mul/fma_f32: one
// VALU issue per two pixels; the per-component arithmetic is that of the scalar formulation).
// HBM: forward reads 8 B and writes 12 B per element (+ halo re-reads served by L2), backward reads
// 20 B and writes 4 B; measured they are VALU/LDS-bound (profiles/).
#include "common.h"

namespace {

constexpr int TS = 16;          // thread block edge (16 x 16 threads) and output tile HEIGHT
constexpr int TSX = 32;         // output tile WIDTH: two columns (lx, lx + 16) per thread, so that a 512^2 x 3 image is 1536
                                // workgroups = 6 per CU = one resident round (16 x 16 tiles were 3072 = 1.5 rounds of 8)
constexpr int HALO = 5;         // window radius
constexpr int TH = TS + 2 * HALO;    // 26 input rows per tile
constexpr int THX = TSX + 2 * HALO;  // 42 input columns per tile
constexpr int SXP = 48;              // staged row stride in floats: the four rows a wave reads at once (16 lanes each) fall into
                                     // disjoint 16-bank groups (48 = 16 mod 32 ... 0, 48, 96, 144 -> banks 0, 48, 32, 16)
constexpr int STAGE_ITERS = (TH * SXP + 255) / 256;  // 5
constexpr int ROW_ITEMS = TH * TS;   // row-pass work items: (staged row, column pair)
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

// normalised Gaussian(11, sigma 1.5) — the coefficients the reference's create_window() produces
__device__ __forceinline__ float gw(int k) {
  constexpr float w[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                           0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
                           0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};
  return w[k];
}

__device__ __forceinline__ float ssim_rcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                   float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                   float* __restrict__ dm_dsigma12, float* __restrict__ partial /*[nblocks,2]*/,
                                                   int crop /* 0: "same" padding; HALO: "valid" — the SSIM map only counts (and only
                                                               passes gradient) where the 11x11 window lies inside the image */) {
  __shared__ float s_x[TH][SXP];
  __shared__ float s_y[TH][SXP];
  __shared__ gs_v2f s_h[5][TH][TS];  // row-pass results, columns (c, c + 16) as one pair
  __shared__ float s_red[2][4];
  const int tid = threadIdx.y * TS + threadIdx.x;
  const int plane = blockIdx.z;
  const int ox = blockIdx.x * TSX, oy = blockIdx.y * TS;
  const float* p1 = img1 + (size_t)plane * H * W;
  const float* p2 = img2 + (size_t)plane * H * W;
  // Staging: every load goes to a clamped (always valid) address and is issued before the first LDS write, so a thread
  // waits for HBM/L2 once — as a loop of predicated loads this was ten dependent round trips per thread and most of the
  // kernel's duration.
  {
    float vx[STAGE_ITERS], vy[STAGE_ITERS];
#pragma unroll
    for (int j = 0; j < STAGE_ITERS; ++j) {
      const int i = tid + 256 * j, r = i / SXP, c = i - r * SXP;
      const size_t o = (size_t)min(max(oy + r - HALO, 0), H - 1) * W + min(max(ox + c - HALO, 0), W - 1);
      vx[j] = p1[o]; vy[j] = p2[o];
    }
#pragma unroll
    for (int j = 0; j < STAGE_ITERS; ++j) {
      const int i = tid + 256 * j, r = i / SXP, c = i - r * SXP;
      const int gy = oy + r - HALO, gx = ox + c - HALO;
      const bool in = c < THX && gy >= 0 && gy < H && gx >= 0 && gx < W;
      if (i < TH * SXP) { s_x[r][c] = in ? vx[j] : 0.f; s_y[r][c] = in ? vy[j] : 0.f; }
    }
  }
  __syncthreads();
  for (int i = tid; i < ROW_ITEMS; i += 256) {
    const int r = i >> 4, c = i & 15;
    gs_v2f sx = {0.f, 0.f}, sy = sx, sxx = sx, syy = sx, sxy = sx;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const gs_v2f w = {gw(k), gw(k)};
      const gs_v2f x = {s_x[r][c + k], s_x[r][c + TS + k]}, y = {s_y[r][c + k], s_y[r][c + TS + k]};
      const gs_v2f wx = w * x, wy = w * y;
      sx = gs_fma2(w, x, sx); sy = gs_fma2(w, y, sy);
      sxx = gs_fma2(wx, x, sxx); syy = gs_fma2(wy, y, syy); sxy = gs_fma2(wx, y, sxy);
    }
    s_h[0][r][c] = sx; s_h[1][r][c] = sy; s_h[2][r][c] = sxx; s_h[3][r][c] = syy; s_h[4][r][c] = sxy;
  }
  __syncthreads();
  const int ly = threadIdx.y, lx = threadIdx.x;
  float val = 0.f, l1 = 0.f;
  {
    gs_v2f mu1 = {0.f, 0.f}, mu2 = mu1, exx = mu1, eyy = mu1, exy = mu1;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const gs_v2f w = {gw(k), gw(k)};
      mu1 = gs_fma2(w, s_h[0][ly + k][lx], mu1); mu2 = gs_fma2(w, s_h[1][ly + k][lx], mu2);
      exx = gs_fma2(w, s_h[2][ly + k][lx], exx); eyy = gs_fma2(w, s_h[3][ly + k][lx], eyy);
      exy = gs_fma2(w, s_h[4][ly + k][lx], exy);
    }
    const gs_v2f two = {2.f, 2.f}, c1 = {C1, C1}, c2 = {C2, C2};
    const gs_v2f mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const gs_v2f s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
    const gs_v2f A = mu1_sq + mu2_sq + c1, B = s1 + s2 + c2, Cc = two * mu12 + c1, Dd = two * s12 + c2;
    // two reciprocals per pixel (v_rcp_f32 + one Newton step: <= 1 ulp) instead of four IEEE divisions (~10 VALU ops each)
    const gs_v2f invA = {ssim_rcp(A[0]), ssim_rcp(A[1])}, invB = {ssim_rcp(B[0]), ssim_rcp(B[1])};
    const gs_v2f invAB = invA * invB;
    const gs_v2f m = Cc * Dd * invAB;
    const gs_v2f t = mu1 * two * Cc * Dd * invAB;
    const gs_v2f d1 = (mu2 * two * Dd) * invAB - (mu2 * two * Cc) * invAB - t * invA + t * invB;
    const gs_v2f d2 = -m * invB;
    const gs_v2f d3 = two * Cc * invAB;
    const int gy = oy + ly;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int gx = ox + lx + half * TS;
      if (gx < W && gy < H) {
        const bool counted = gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;
        val += counted ? m[half] : 0.f;
        const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
        if (dm_dmu1) {
          dm_dmu1[o] = counted ? d1[half] : 0.f; dm_dsigma1_sq[o] = counted ? d2[half] : 0.f; dm_dsigma12[o] = counted ? d3[half] : 0.f;
        }
        l1 += fabsf(s_x[ly + HALO][lx + half * TS + HALO] - s_y[ly + HALO][lx + half * TS + HALO]);
      }
    }
  }
  val = gs_wave_sum_row3(val);   // six DPP adds each (a shuffle butterfly is six LDS-crossbar round trips); the total is in lane 63
  l1 = gs_wave_sum_row3(l1);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { s_red[0][wave] = val; s_red[1][wave] = l1; }
  __syncthreads();
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    partial[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
  }
}

// deterministic final reduction of the per-block partial sums (double accumulation)
__global__ __launch_bounds__(1024) void k_ssim_finish(int nblocks, double inv_n, const float* __restrict__ partial,
                                                       float* __restrict__ ssim_mean, float* __restrict__ l1_mean,
                                                       float* __restrict__ loss, float lambda_dssim) {
  __shared__ double s_a[16], s_b[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 1024) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
  a = gs_wave_sum_row3_f64(a); b = gs_wave_sum_row3_f64(b);   // DPP adds; the totals are in lane 63
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63) { s_a[wave] = a; s_b[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) { ta += s_a[w]; tb += s_b[w]; }
    const float sm = (float)(ta * inv_n), lm = (float)(tb * inv_n);
    if (ssim_mean) *ssim_mean = sm;
    if (l1_mean) *l1_mean = lm;
    if (loss) *loss = (1.0f - lambda_dssim) * lm + lambda_dssim * (1.0f - sm);  // reference train.py:176
  }
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, float inv_n, const float* __restrict__ img1, const float* __restrict__ img2,
                                                   const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                   const float* __restrict__ dm_dsigma12, const float* __restrict__ ssim_scale,
                                                   const float* __restrict__ l1_scale, float ssim_scale_host, float l1_scale_host,
                                                   float* __restrict__ dL_dimg1) {
  __shared__ float s_a[TH][SXP];
  __shared__ float s_b[TH][SXP];
  __shared__ float s_c[TH][SXP];
  __shared__ gs_v2f s_h[3][TH][TS];  // columns (c, c + 16) as one pair, see k_ssim_fwd
  const int tid = threadIdx.y * TS + threadIdx.x;
  const int plane = blockIdx.z;
  const int ox = blockIdx.x * TSX, oy = blockIdx.y * TS;
  const size_t po = (size_t)plane * H * W;
  // scale = (device scalar, if given) x (host scalar)
  const float ks = (ssim_scale ? *ssim_scale : 1.f) * ssim_scale_host * inv_n;
  const float kl = (l1_scale ? *l1_scale : 1.f) * l1_scale_host * inv_n;
  // the two output pixels' own values: requested now, used after the passes
  float px[2], py[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const size_t o = po + (size_t)min(oy + (int)threadIdx.y, H - 1) * W + min(ox + (int)threadIdx.x + half * TS, W - 1);
    px[half] = img1[o]; py[half] = img2[o];
  }
  if (ks != 0.f) {
    {  // staged with clamped addresses, all loads in flight before the first LDS write (see k_ssim_fwd)
      float va[STAGE_ITERS], vb[STAGE_ITERS], vc[STAGE_ITERS];
#pragma unroll
      for (int j = 0; j < STAGE_ITERS; ++j) {
        const int i = tid + 256 * j, r = i / SXP, c = i - r * SXP;
        const size_t o = po + (size_t)min(max(oy + r - HALO, 0), H - 1) * W + min(max(ox + c - HALO, 0), W - 1);
        va[j] = dm_dmu1[o]; vb[j] = dm_dsigma1_sq[o]; vc[j] = dm_dsigma12[o];
      }
#pragma unroll
      for (int j = 0; j < STAGE_ITERS; ++j) {
        const int i = tid + 256 * j, r = i / SXP, c = i - r * SXP;
        const int gy = oy + r - HALO, gx = ox + c - HALO;
        const bool in = c < THX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (i < TH * SXP) { s_a[r][c] = in ? va[j] : 0.f; s_b[r][c] = in ? vb[j] : 0.f; s_c[r][c] = in ? vc[j] : 0.f; }
      }
    }
    __syncthreads();
    for (int i = tid; i < ROW_ITEMS; i += 256) {
      const int r = i >> 4, c = i & 15;
      gs_v2f a = {0.f, 0.f}, b = a, cc = a;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const gs_v2f w = {gw(k), gw(k)};
        a = gs_fma2(w, gs_v2f{s_a[r][c + k], s_a[r][c + TS + k]}, a);
        b = gs_fma2(w, gs_v2f{s_b[r][c + k], s_b[r][c + TS + k]}, b);
        cc = gs_fma2(w, gs_v2f{s_c[r][c + k], s_c[r][c + TS + k]}, cc);
      }
      s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = cc;
    }
    __syncthreads();
  }
  const int ly = threadIdx.y, lx = threadIdx.x;
  const int gy = oy + ly;
  if (gy >= H) return;
  gs_v2f a = {0.f, 0.f}, b = a, cc = a;
  if (ks != 0.f) {
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const gs_v2f w = {gw(k), gw(k)};
      a = gs_fma2(w, s_h[0][ly + k][lx], a); b = gs_fma2(w, s_h[1][ly + k][lx], b); cc = gs_fma2(w, s_h[2][ly + k][lx], cc);
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int gx = ox + lx + half * TS;
    if (gx >= W) continue;
    const size_t o = po + (size_t)gy * W + gx;
    const float x = px[half], y = py[half];
    float g = 0.f;
    if (ks != 0.f) g = ks * (a[half] + 2.f * x * b[half] + y * cc[half]);
    if (kl != 0.f) {
      const float d = x - y;
      g += kl * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    dL_dimg1[o] = g;
  }
}


// ------------------------------------------------------------------------------------------------
// The training loss and its gradient in ONE pass: loss = (1-l) * L1 + l * (1 - SSIM) (reference train.py:171-176) and
// dloss/dimg1.  The two-kernel formulation above writes the three SSIM partial-derivative maps (36 B per pixel-channel) to
// HBM and reads them back with a 2.1x halo amplification — 75 % of its traffic (profiles/r02_pmc_c4_*: 193 + 228 MB per 1080p
// frame).  Here a workgroup owns a 32x32 output tile and recomputes the forward on the tile + window radius (42x42) from
// inputs staged over tile + 2 radii (52x52); the maps only ever exist in LDS.
//   A  stage x, y over 52x52 (zero outside the image)                                   -> s_x, s_y
//   B  forward row pass: rows 0..51, columns 0..41, five moments                        -> s_h
//   C  forward column pass + SSIM formula on 42x42: value (tile pixels only), d1..d3     -> s_d (over s_x / s_y), zero outside the image
//   D  backward row pass on d1..d3: rows 0..41, columns 0..31                           -> s_e (over s_h)
//   E  backward column pass on the 32x32 tile + gradient + L1 term                      -> global
// Every pass is a register sliding window: a thread produces a run of 6 / 7 / 8 / 4 consecutive outputs along the pass
// direction from run + 10 inputs read once (the per-tap formulation read each input 11 times from LDS).
constexpr int FT = 32;             // fused output tile edge
constexpr int FR1 = FT + 2 * HALO;  // 42
constexpr int FR2 = FT + 4 * HALO;  // 52
// LDS pitches (floats), chosen with the lane -> item mappings below so that no pass has bank conflicts: a row pass reads pairs
// (8-byte loads) with the ROW index varying fastest across lanes, which is conflict-free when the pitch is 2 x odd; a column pass
// and every store walk consecutive columns across lanes, conflict-free for any pitch, and an odd pitch keeps the row passes'
// scattered 4-byte stores (consecutive rows across lanes) on distinct banks too.  (First version: pitches 56 / 44 / 32 with the
// column group varying fastest — the backward row pass's stores were 16-way conflicts.)
constexpr int FXP = 58;            // staged inputs (52 wide), read in pairs by the forward row pass
constexpr int FHP = 43;            // row-pass moments (42 wide), stored one float at a time, read down columns
constexpr int FDP = 46;            // d1..d3 (42 wide), read in pairs by the backward row pass
constexpr int FEP = 33;            // backward row-pass results (32 wide), stored one float at a time, read down columns
constexpr int FTHREADS = 512;      // eight waves per workgroup: the tile's LDS (69 KB) allows two workgroups per CU, so this is
                                   // what puts four waves on every SIMD
constexpr int F_STAGE_ITERS = (FR2 * FXP + FTHREADS - 1) / FTHREADS;   // 6
constexpr int FER = FT * FT / FTHREADS;   // output rows per thread in the last pass (2)
// Run lengths of the column pass C and the row pass D (outputs per item).  A longer run reads fewer inputs per output (run + 10
// for run outputs) but makes fewer items: 7 / 8 give 252 / 168 items for 512 threads, i.e. phase C runs on four of the
// workgroup's eight waves.  Shorter runs that fill the workgroup were measured and are slower at 512^2 (A/B build switches,
// profiles/r03_ab_ssim_run_lengths_kernel_avg.txt): C/D = 7/8 17.2 us (shipped), 4/8 17.5, 4/4 18.0, 6/4 18.9 — the second
// resident workgroup of the CU already fills the SIMDs, and the extra LDS reads per output cost more than the idle waves.
#ifndef GS_SSIM_RUN_C
#define GS_SSIM_RUN_C 7
#endif
#ifndef GS_SSIM_RUN_D
#define GS_SSIM_RUN_D 8
#endif
constexpr int FCR = GS_SSIM_RUN_C, FC_GROUPS = (FR1 + FCR - 1) / FCR;   // 7 -> 6 groups x 42 columns = 252 items; 4 -> 11 x 42 = 462
constexpr int FDR = GS_SSIM_RUN_D, FD_GROUPS = FT / FDR;                // 8 -> 4 groups x 42 rows = 168 items; 4 -> 8 x 42 = 336
static_assert(FC_GROUPS * FR1 <= FTHREADS && FD_GROUPS * FR1 <= FTHREADS && FT % FDR == 0 && FDR % 2 == 0, "one item per thread");

__global__ __launch_bounds__(FTHREADS) void k_l1_ssim_fused(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                        float ks /* d loss / d ssim_mean / N */, float kl /* d loss / d l1_mean / N */,
                                                        float* __restrict__ dL_dimg1, float* __restrict__ partial /*[nblocks,2]*/) {
  __shared__ float s_xy[2][FR2][FXP];        // x, y staged; later d1..d3 [3][FR1][FDP] (5796 <= 6032 floats)
  __shared__ float s_h[5][FR2][FHP];         // row-pass moments; later the backward row pass [3][FR1][FEP]
  static_assert(3 * FR1 * FDP <= 2 * FR2 * FXP && 3 * FR1 * FEP <= 5 * FR2 * FHP, "aliased LDS buffers must fit");
  __shared__ float s_red[2][FTHREADS / 64];
  float (*s_d)[FR1][FDP] = reinterpret_cast<float (*)[FR1][FDP]>(&s_xy[0][0][0]);
  float (*s_e)[FR1][FEP] = reinterpret_cast<float (*)[FR1][FEP]>(&s_h[0][0][0]);
  const int tid = threadIdx.x;
  const int plane = blockIdx.z;
  const int ox = blockIdx.x * FT, oy = blockIdx.y * FT;
  const size_t po = (size_t)plane * H * W;
  const float* p1 = img1 + po;
  const float* p2 = img2 + po;
  // the thread's own output pixels (phase E: column ec, rows er0..er0+FER-1), requested first, used last
  const int ec = tid & 31, er0 = (tid >> 5) * FER;
  float ex[FER], ey[FER];
#pragma unroll
  for (int j = 0; j < FER; ++j) {
    const size_t o = (size_t)min(oy + er0 + j, H - 1) * W + min(ox + ec, W - 1);
    ex[j] = p1[o]; ey[j] = p2[o];
  }
  // ---- A: staging, every load to a clamped (valid) address and in flight before the first LDS write
  {
    float vx[F_STAGE_ITERS], vy[F_STAGE_ITERS];
#pragma unroll
    for (int j = 0; j < F_STAGE_ITERS; ++j) {
      const int i = tid + FTHREADS * j, r = i / FXP, c = i - r * FXP;
      const size_t o = (size_t)min(max(oy + r - 2 * HALO, 0), H - 1) * W + min(max(ox + c - 2 * HALO, 0), W - 1);
      vx[j] = p1[o]; vy[j] = p2[o];
    }
#pragma unroll
    for (int j = 0; j < F_STAGE_ITERS; ++j) {
      const int i = tid + FTHREADS * j, r = i / FXP, c = i - r * FXP;
      const int gy = oy + r - 2 * HALO, gx = ox + c - 2 * HALO;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      if (i < FR2 * FXP) { s_xy[0][r][c] = in ? vx[j] : 0.f; s_xy[1][r][c] = in ? vy[j] : 0.f; }
    }
  }
  __syncthreads();
  // ---- B: forward row pass, item = (row, 6 consecutive columns): 16 inputs of x and y -> 6 x 5 outputs
  for (int i = tid; i < FR2 * 7; i += FTHREADS) {
    const int r = i % FR2, c0 = (i / FR2) * 6;   // rows vary fastest across lanes
    float x[16], y[16];
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const float2 a = *reinterpret_cast<const float2*>(&s_xy[0][r][c0 + t]);
      const float2 b = *reinterpret_cast<const float2*>(&s_xy[1][r][c0 + t]);
      x[t] = a.x; x[t + 1] = a.y; y[t] = b.x; y[t + 1] = b.y;
    }
    float sx[6], sy[6], sxx[6], syy[6], sxy[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { sx[j] = sy[j] = sxx[j] = syy[j] = sxy[j] = 0.f; }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float xx = x[t] * x[t], yy = y[t] * y[t], xy = x[t] * y[t];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int k = t - j;   // tap index of input t for output j
        if (k >= 0 && k < 11) {
          const float w = gw(k);
          sx[j] = fmaf(w, x[t], sx[j]); sy[j] = fmaf(w, y[t], sy[j]);
          sxx[j] = fmaf(w, xx, sxx[j]); syy[j] = fmaf(w, yy, syy[j]); sxy[j] = fmaf(w, xy, sxy[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      s_h[0][r][c0 + j] = sx[j]; s_h[1][r][c0 + j] = sy[j]; s_h[2][r][c0 + j] = sxx[j]; s_h[3][r][c0 + j] = syy[j];
      s_h[4][r][c0 + j] = sxy[j];
    }
  }
  __syncthreads();
  // ---- C: forward column pass + SSIM formula, item = (column, 7 consecutive rows): 17 inputs per moment -> 7 outputs
  float val = 0.f;
  if (tid < FR1 * FC_GROUPS) {
    const int c = tid % FR1, r0 = (tid / FR1) * FCR;
    float mo[5][FCR];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      float v[FCR + 10];
#pragma unroll
      for (int t = 0; t < FCR + 10; ++t) v[t] = s_h[m][min(r0 + t, FR2 - 1)][c];   // (rows past the last one feed outputs that are not kept)
#pragma unroll
      for (int j = 0; j < FCR; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a = fmaf(gw(k), v[j + k], a);
        mo[m][j] = a;
      }
    }
    const int gx = ox + c - HALO;
#pragma unroll
    for (int j = 0; j < FCR; ++j) {
      const int r = r0 + j, gy = oy + r - HALO;
      const float mu1 = mo[0][j], mu2 = mo[1][j];
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = mo[2][j] - mu1_sq, s2 = mo[3][j] - mu2_sq, s12 = mo[4][j] - mu12;
      const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2, Cc = 2.f * mu12 + C1, Dd = 2.f * s12 + C2;
      const float invA = ssim_rcp(A), invB = ssim_rcp(B), invAB = invA * invB;
      const float m = Cc * Dd * invAB;
      const float t = mu1 * 2.f * Cc * Dd * invAB;
      const float d1 = (mu2 * 2.f * Dd) * invAB - (mu2 * 2.f * Cc) * invAB - t * invA + t * invB;
      const float d2 = -m * invB;
      const float d3 = 2.f * Cc * invAB;
      const bool in_img = gx >= 0 && gx < W && gy >= 0 && gy < H && r < FR1;
      // (the staged inputs stay readable until the barrier below: s_d aliases them, and phase C reads only s_h)
      mo[0][j] = in_img ? d1 : 0.f; mo[1][j] = in_img ? d2 : 0.f; mo[2][j] = in_img ? d3 : 0.f;
      const bool in_tile = c >= HALO && c < HALO + FT && r >= HALO && r < HALO + FT;
      val += (in_img && in_tile) ? m : 0.f;
    }
    // all phase-B readers of s_xy are past the barrier above, so the maps may overwrite it
#pragma unroll
    for (int j = 0; j < FCR; ++j)
      if (FR1 % FCR == 0 || r0 + j < FR1) { s_d[0][r0 + j][c] = mo[0][j]; s_d[1][r0 + j][c] = mo[1][j]; s_d[2][r0 + j][c] = mo[2][j]; }
  }
  __syncthreads();
  // ---- D: backward row pass, item = (row, 8 consecutive columns): 18 inputs per map -> 8 outputs
  if (tid < FR1 * FD_GROUPS) {
    const int r = tid % FR1, c0 = (tid / FR1) * FDR;   // rows vary fastest across lanes
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float v[FDR + 10];
#pragma unroll
      for (int t = 0; t < FDR + 10; t += 2) {
        const float2 a = *reinterpret_cast<const float2*>(&s_d[m][r][c0 + t]);
        v[t] = a.x; v[t + 1] = a.y;
      }
#pragma unroll
      for (int j = 0; j < FDR; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a = fmaf(gw(k), v[j + k], a);
        s_e[m][r][c0 + j] = a;   // s_e aliases s_h: every phase-C reader of s_h is past the barrier above
      }
    }
  }
  __syncthreads();
  // ---- E: backward column pass on the tile, item = (column, FER rows): FER + 10 inputs per map -> FER outputs; gradient + L1
  float l1 = 0.f;
  {
    float out[3][FER];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float v[FER + 10];
#pragma unroll
      for (int t = 0; t < FER + 10; ++t) v[t] = s_e[m][er0 + t][ec];
#pragma unroll
      for (int j = 0; j < FER; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a = fmaf(gw(k), v[j + k], a);
        out[m][j] = a;
      }
    }
    const int gx = ox + ec;
#pragma unroll
    for (int j = 0; j < FER; ++j) {
      const int gy = oy + er0 + j;
      if (gx < W && gy < H) {
        const float x = ex[j], y = ey[j], d = x - y;
        l1 += fabsf(d);
        const float g = ks * (out[0][j] + 2.f * x * out[1][j] + y * out[2][j]) + kl * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        dL_dimg1[po + (size_t)gy * W + gx] = g;
      }
    }
  }
  val = gs_wave_sum_row3(val);   // six DPP adds each (a shuffle butterfly is six LDS-crossbar round trips); the total is in lane 63
  l1 = gs_wave_sum_row3(l1);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) { s_red[0][wave] = val; s_red[1][wave] = l1; }
  __syncthreads();
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < FTHREADS / 64; ++w) { ta += s_red[0][w]; tb += s_red[1][w]; }
    partial[2 * b] = ta;
    partial[2 * b + 1] = tb;
  }
}

// ---- l1_loss(network_output, gt) = mean |a - b| (reference utils/loss_utils.py:39-40; train.py:171) as one pass + the
// finishing launch above instead of PyTorch's sub / abs / mean, and its gradient sgn(a - b) * g / n as one launch instead of
// the four of (mean, abs, sub)'s backward nodes.  Partial sums per workgroup in float, finished in double, in a fixed order.
constexpr int L1_THREADS = 256, L1_PER_THREAD = 16;   // 4096 elements per workgroup
__global__ __launch_bounds__(L1_THREADS) void k_l1_partial(long long n, const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ partial) {
  __shared__ float s_red[L1_THREADS / 64];
  const long long base = (long long)blockIdx.x * (L1_THREADS * L1_PER_THREAD);
  float acc = 0.f;
  const bool vec = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) && base + L1_THREADS * L1_PER_THREAD <= n;
  if (vec) {
#pragma unroll
    for (int r = 0; r < L1_PER_THREAD / 4; ++r) {
      const long long j = (base >> 2) + threadIdx.x + (long long)L1_THREADS * r;
      const float4 x = reinterpret_cast<const float4*>(a)[j], y = reinterpret_cast<const float4*>(b)[j];
      acc += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
    }
  } else {
    for (int r = 0; r < L1_PER_THREAD; ++r) {
      const long long i = base + threadIdx.x + (long long)L1_THREADS * r;
      if (i < n) acc += fabsf(a[i] - b[i]);
    }
  }
  acc = gs_wave_sum_row3(acc);   // the total is in lane 63
  if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    partial[2 * blockIdx.x + 1] = 0.f;   // (the finishing kernel sums pairs: [ssim, l1] for its other callers)
  }
}

__global__ __launch_bounds__(256) void k_l1_bwd(long long n, const float* __restrict__ a, const float* __restrict__ b,
                                                const float* __restrict__ grad_scale, float n_as_float, float* __restrict__ d_a) {
  // what autograd computes for abs(a - b).mean(): the incoming gradient divided by the element count, times sgn(a - b) (0 at 0)
  const float s = *grad_scale / n_as_float;
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (((((uintptr_t)a | (uintptr_t)b | (uintptr_t)d_a) & 15) == 0) && i0 + 4 <= n) {
    const float4 x = reinterpret_cast<const float4*>(a)[i0 >> 2], y = reinterpret_cast<const float4*>(b)[i0 >> 2];
    auto sg = [&](float d) { return d > 0.f ? s : (d < 0.f ? -s : 0.f); };
    reinterpret_cast<float4*>(d_a)[i0 >> 2] = make_float4(sg(x.x - y.x), sg(x.y - y.y), sg(x.z - y.z), sg(x.w - y.w));
  } else {
    for (long long i = i0; i < min(n, i0 + 4); ++i) { const float d = a[i] - b[i]; d_a[i] = d > 0.f ? s : (d < 0.f ? -s : 0.f); }
  }
}

// ---- the reference's loss EXPRESSION as written (train.py:171-176): l1_loss(image, gt), fused_ssim(image, gt) and scalar
// arithmetic between two 0-dim tensors.  mi355gs_l1_ssim_pair_forward leaves both means and d(ssim_mean)/dimg1 from the one
// pass of k_l1_ssim_fused (ks = 1/N, kl = 0); the scalar arithmetic the host has recorded is evaluated by k_loss_program, and
// the backward of the whole expression is ONE launch of k_loss_pair_bwd over the image:
//   d_img1 = (gl * c_l1 / n) * sgn(a - b) + (gs * c_ssim) * dssim      (gl, gs: incoming gradients, device scalars or 1)
// in autograd's own operation order for abs(a - b).mean() and for a scaled saved gradient.
struct GsLossProgram {
  int n;
  signed char op[MI355GS_LOSS_PROGRAM_MAX];
  float k[MI355GS_LOSS_PROGRAM_MAX];
};

// One launch: the two means finished from the pass's per-workgroup partial sums (the reduction of k_ssim_finish: double
// accumulation, fixed order — the same bits), left in *ssim_mean / *l1_mean, then the recorded program evaluated on them.
__device__ __forceinline__ void loss_program_body(const GsLossProgram& p, int nblocks, double inv_n, const float* __restrict__ partial,
                                                  float* __restrict__ ssim_mean, float* __restrict__ l1_mean, float* __restrict__ out,
                                                  float* __restrict__ host_out, float ticket) {
#pragma clang fp contract(off)   // one rounding per recorded operation, as eager PyTorch's elementwise kernels
  __shared__ double s_a[16], s_b[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 1024) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
  a = gs_wave_sum_row3_f64(a); b = gs_wave_sum_row3_f64(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63) { s_a[wave] = a; s_b[wave] = b; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double ta = 0.0, tb = 0.0;
  for (int w = 0; w < 16; ++w) { ta += s_a[w]; tb += s_b[w]; }
  const float sm = (float)(ta * inv_n), lm = (float)(tb * inv_n);
  *ssim_mean = sm; *l1_mean = lm;
  float st[MI355GS_LOSS_PROGRAM_MAX];
  int sp = 0;
  for (int i = 0; i < p.n; ++i) {
    const float k = p.k[i];
    switch (p.op[i]) {
      case MI355GS_LOSS_OP_L1: st[sp++] = lm; break;
      case MI355GS_LOSS_OP_SSIM: st[sp++] = sm; break;
      case MI355GS_LOSS_OP_MULK: st[sp - 1] = st[sp - 1] * k; break;
      case MI355GS_LOSS_OP_ADDK: st[sp - 1] = st[sp - 1] + k; break;
      case MI355GS_LOSS_OP_RSUBK: st[sp - 1] = k - st[sp - 1]; break;
      case MI355GS_LOSS_OP_DIVK: st[sp - 1] = st[sp - 1] * (1.0f / k); break;   // a tensor divided by a host scalar: x * (1 / k), as ATen's kernel
      case MI355GS_LOSS_OP_NEG: st[sp - 1] = -st[sp - 1]; break;
      case MI355GS_LOSS_OP_ADD: st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; break;
      case MI355GS_LOSS_OP_SUB: st[sp - 2] = st[sp - 2] - st[sp - 1]; --sp; break;
      default: break;
    }
  }
  const float value = sp > 0 ? st[sp - 1] : 0.f;
  *out = value;
  // ... and, if asked, straight to the host: (value, ticket) as ONE 8-byte store into pinned, device-mapped memory — whoever
  // polls the ticket has the value without a copy, a stream synchronisation or a wait for the kernels enqueued behind this one
  if (host_out) {
    typedef float v2f __attribute__((vector_size(8)));
    v2f both = {value, ticket};
    *reinterpret_cast<v2f*>(host_out) = both;
  }
}

__global__ __launch_bounds__(1024) void k_loss_program(GsLossProgram p, int nblocks, double inv_n, const float* __restrict__ partial,
                                                      float* __restrict__ ssim_mean, float* __restrict__ l1_mean, float* __restrict__ out,
                                                      float* __restrict__ host_out, float ticket) {
  loss_program_body(p, nblocks, inv_n, partial, ssim_mean, l1_mean, out, host_out, ticket);
}

// elements [i0, i0 + 4) of d_a = sl * sgn(a - b) + ss * dssim
__device__ __forceinline__ void loss_pair_bwd_four(long long n, long long i0, const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ dssim, float sl, float ss, bool use_l1, bool use_ss,
                                                   float* __restrict__ d_a) {
#pragma clang fp contract(off)
  auto one = [&](float x, float y, float ds) {
    const float d = x - y;
    const float tl = use_l1 ? (d > 0.f ? sl : (d < 0.f ? -sl : 0.f)) : 0.f;
    return use_ss ? tl + ss * ds : tl;
  };
  const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)d_a | (uintptr_t)(use_ss ? dssim : a);
  if (((al & 15) == 0) && i0 + 4 <= n) {
    const float4 x = reinterpret_cast<const float4*>(a)[i0 >> 2], y = reinterpret_cast<const float4*>(b)[i0 >> 2];
    const float4 s = use_ss ? reinterpret_cast<const float4*>(dssim)[i0 >> 2] : make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(d_a)[i0 >> 2] = make_float4(one(x.x, y.x, s.x), one(x.y, y.y, s.y), one(x.z, y.z, s.z), one(x.w, y.w, s.w));
  } else {
    for (long long i = i0; i < min(n, i0 + 4); ++i) d_a[i] = one(a[i], b[i], use_ss ? dssim[i] : 0.f);
  }
}

__global__ __launch_bounds__(256) void k_loss_pair_bwd(long long n, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ dssim, const float* __restrict__ g_l1, float c_l1,
                                                       const float* __restrict__ g_ssim, float c_ssim, float n_as_float,
                                                       float* __restrict__ d_a) {
#pragma clang fp contract(off)
  const float sl = ((g_l1 ? *g_l1 : 1.0f) * c_l1) / n_as_float;
  const float ss = (g_ssim ? *g_ssim : 1.0f) * c_ssim;
  const bool use_l1 = c_l1 != 0.0f, use_ss = c_ssim != 0.0f && dssim != nullptr;   // (a term that was never asked for adds no 0 * inf)
  loss_pair_bwd_four(n, ((long long)blockIdx.x * 256 + threadIdx.x) * 4, a, b, dssim, sl, ss, use_l1, use_ss, d_a);
}

// loss.backward() right behind the materialisation, i.e. with dL/d(value) = 1 known before anything is launched: ONE launch for
// the value (workgroup 0: the means finished, the program evaluated — loss_program_body) and for the gradient of the recorded
// expression over the image (the other workgroups: k_loss_pair_bwd's arithmetic with g = 1, which does not depend on the means)
__global__ __launch_bounds__(1024) void k_loss_program_grad(GsLossProgram p, int nblocks, double inv_n, const float* __restrict__ partial,
                                                           float* __restrict__ ssim_mean, float* __restrict__ l1_mean, float* __restrict__ out,
                                                           float* __restrict__ host_out, float ticket, long long n,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ dssim, float c_l1, float c_ssim, float n_as_float,
                                                           float* __restrict__ d_a) {
#pragma clang fp contract(off)
  if (blockIdx.x == 0) {
    loss_program_body(p, nblocks, inv_n, partial, ssim_mean, l1_mean, out, host_out, ticket);
    return;
  }
  const float sl = (1.0f * c_l1) / n_as_float;
  const float ss = 1.0f * c_ssim;
  const bool use_l1 = c_l1 != 0.0f, use_ss = c_ssim != 0.0f && dssim != nullptr;
  loss_pair_bwd_four(n, ((long long)(blockIdx.x - 1) * 1024 + threadIdx.x) * 4, a, b, dssim, sl, ss, use_l1, use_ss, d_a);
}

}  // namespace

static inline int l1_nblocks(long long n) { return (int)((n + L1_THREADS * L1_PER_THREAD - 1) / (L1_THREADS * L1_PER_THREAD)); }

static inline int ssim_nblocks(int B, int C, int H, int W) { return B * C * ((H + TS - 1) / TS) * ((W + TSX - 1) / TSX); }

extern "C" {

size_t mi355gs_ssim_scratch_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 256;
  return gs_align((size_t)ssim_nblocks(B, C, H, W) * 2 * sizeof(float));
}

int mi355gs_ssim_forward(void* stream_, int B, int C, int H, int W, const float* img1, const float* img2, float* dm_dmu1,
                         float* dm_dsigma1_sq, float* dm_dsigma12, void* scratch, float* ssim_mean, float* l1_mean, int padding_valid) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !scratch) return MI355GS_EINVAL;
  const int crop = padding_valid ? HALO : 0;
  if (padding_valid && (l1_mean || H <= 2 * HALO || W <= 2 * HALO)) return MI355GS_EINVAL;
  if ((dm_dmu1 == nullptr) != (dm_dsigma1_sq == nullptr) || (dm_dmu1 == nullptr) != (dm_dsigma12 == nullptr)) return MI355GS_EINVAL;
  if ((size_t)B * C > 65535) return MI355GS_EINVAL;
  const dim3 grid((W + TSX - 1) / TSX, (H + TS - 1) / TS, B * C);
  GS_KRANGE("ssim_fwd");
  hipLaunchKernelGGL(k_ssim_fwd, grid, dim3(TS, TS), 0, stream, H, W, img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (float*)scratch, crop);
  GS_CHECK_LAUNCH("ssim_fwd");
  // one finishing launch normalises both sums by the same count: with "valid" padding that is the cropped map's, so the L1
  // mean (a "same"-padding quantity of the training loss) is only offered with padding_valid == 0
  const double inv_n = 1.0 / ((double)B * C * (H - 2 * crop) * (W - 2 * crop));
  GS_KRANGE("ssim_finish");
  hipLaunchKernelGGL(k_ssim_finish, dim3(1), dim3(1024), 0, stream, ssim_nblocks(B, C, H, W), inv_n, (const float*)scratch, ssim_mean,
                     l1_mean, (float*)nullptr, 0.f);
  GS_CHECK_LAUNCH("ssim_finish");
  return MI355GS_OK;
}

int mi355gs_ssim_backward(void* stream_, int B, int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                          const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* ssim_grad_scale,
                          const float* l1_grad_scale, float* dL_dimg1, int padding_valid) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dL_dimg1) return MI355GS_EINVAL;
  if (ssim_grad_scale && (!dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12)) return MI355GS_EINVAL;
  if ((size_t)B * C > 65535) return MI355GS_EINVAL;
  const dim3 grid((W + TSX - 1) / TSX, (H + TS - 1) / TS, B * C);
  const int crop = padding_valid ? HALO : 0;
  if (padding_valid && (l1_grad_scale || H <= 2 * HALO || W <= 2 * HALO)) return MI355GS_EINVAL;
  // the forward zeroed the saved partials outside the counted region, so the same kernel serves both paddings
  const float inv_n = (float)(1.0 / ((double)B * C * (H - 2 * crop) * (W - 2 * crop)));
  GS_KRANGE("ssim_bwd");
  hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(TS, TS), 0, stream, H, W, inv_n, img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                     ssim_grad_scale, l1_grad_scale, ssim_grad_scale ? 1.f : 0.f, l1_grad_scale ? 1.f : 0.f, dL_dimg1);
  GS_CHECK_LAUNCH("ssim_bwd");
  return MI355GS_OK;
}

size_t mi355gs_l1_scratch_bytes(int64_t n) { return gs_align((size_t)(n > 0 ? l1_nblocks(n) : 1) * 2 * sizeof(float)); }

int mi355gs_l1_loss_forward(void* stream_, int64_t n, const float* a, const float* b, void* scratch, float* mean_out) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (n <= 0 || n > (int64_t)1 << 40 || !a || !b || !scratch || !mean_out) return MI355GS_EINVAL;
  GS_KRANGE("l1_partial");
  hipLaunchKernelGGL(k_l1_partial, dim3(l1_nblocks(n)), dim3(L1_THREADS), 0, stream, (long long)n, a, b, (float*)scratch);
  GS_CHECK_LAUNCH("l1_partial");
  GS_KRANGE("l1_finish");
  hipLaunchKernelGGL(k_ssim_finish, dim3(1), dim3(1024), 0, stream, l1_nblocks(n), 1.0 / (double)n, (const float*)scratch, mean_out,
                     (float*)nullptr, (float*)nullptr, 0.f);
  GS_CHECK_LAUNCH("l1_finish");
  return MI355GS_OK;
}

int mi355gs_l1_loss_backward(void* stream_, int64_t n, const float* a, const float* b, const float* grad_scale, float* d_a) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (n <= 0 || n > (int64_t)1 << 40 || !a || !b || !grad_scale || !d_a) return MI355GS_EINVAL;
  GS_KRANGE("l1_bwd");
  hipLaunchKernelGGL(k_l1_bwd, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (long long)n, a, b, grad_scale, (float)n, d_a);
  GS_CHECK_LAUNCH("l1_bwd");
  return MI355GS_OK;
}

static inline int fused_nblocks(int B, int C, int H, int W) { return B * C * ((H + FT - 1) / FT) * ((W + FT - 1) / FT); }

// loss = (1-l) * L1 + l * (1 - SSIM) AND dloss/dimg1 from one pass over the images (k_l1_ssim_fused): the binding's forward keeps
// the gradient and its backward only scales it by the incoming dL/dloss.
int mi355gs_l1_ssim_loss_fused(void* stream_, int B, int C, int H, int W, const float* img1, const float* img2, void* scratch,
                               float lambda_dssim, float* ssim_mean, float* l1_mean, float* loss, float* dloss_dimg1) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !scratch || !loss || !dloss_dimg1) return MI355GS_EINVAL;
  if ((size_t)B * C > 65535) return MI355GS_EINVAL;
  const double inv_n = 1.0 / ((double)B * C * H * W);
  const dim3 grid((W + FT - 1) / FT, (H + FT - 1) / FT, B * C);
  {
    GsProfScope prof(3, stream);
    GS_KRANGE("l1_ssim_fused");
    hipLaunchKernelGGL(k_l1_ssim_fused, grid, dim3(FTHREADS), 0, stream, H, W, img1, img2, (float)(-(double)lambda_dssim * inv_n),
                       (float)((1.0 - (double)lambda_dssim) * inv_n), dloss_dimg1, (float*)scratch);
  }
  GS_CHECK_LAUNCH("l1_ssim_fused");
  GS_KRANGE("ssim_finish");
  hipLaunchKernelGGL(k_ssim_finish, dim3(1), dim3(1024), 0, stream, fused_nblocks(B, C, H, W), inv_n, (const float*)scratch, ssim_mean,
                     l1_mean, loss, lambda_dssim);
  GS_CHECK_LAUNCH("ssim_finish");
  return MI355GS_OK;
}

int mi355gs_l1_ssim_pair_forward(void* stream_, int B, int C, int H, int W, const float* img1, const float* img2, void* scratch,
                                 float* dssim_dimg1) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !scratch || !dssim_dimg1) return MI355GS_EINVAL;
  if ((size_t)B * C > 65535) return MI355GS_EINVAL;
  const double inv_n = 1.0 / ((double)B * C * H * W);
  const dim3 grid((W + FT - 1) / FT, (H + FT - 1) / FT, B * C);
  {
    GsProfScope prof(3, stream);
    GS_KRANGE("l1_ssim_pair");
    hipLaunchKernelGGL(k_l1_ssim_fused, grid, dim3(FTHREADS), 0, stream, H, W, img1, img2, (float)inv_n, 0.0f, dssim_dimg1, (float*)scratch);
  }
  GS_CHECK_LAUNCH("l1_ssim_pair");
  return MI355GS_OK;
}

int mi355gs_l1_ssim_pair_backward(void* stream_, int64_t n, const float* img1, const float* img2, const float* dssim_dimg1,
                                  const float* g_l1, float c_l1, const float* g_ssim, float c_ssim, float* d_img1) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (n <= 0 || n > (int64_t)1 << 40 || !img1 || !img2 || !d_img1) return MI355GS_EINVAL;
  if (c_ssim != 0.0f && !dssim_dimg1) return MI355GS_EINVAL;
  GS_KRANGE("loss_pair_bwd");
  hipLaunchKernelGGL(k_loss_pair_bwd, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (long long)n, img1, img2, dssim_dimg1, g_l1,
                     c_l1, g_ssim, c_ssim, (float)n, d_img1);
  GS_CHECK_LAUNCH("loss_pair_bwd");
  return MI355GS_OK;
}

static int build_loss_program(int n_ops, const int32_t* ops, const float* consts, GsLossProgram& p) {
  if (n_ops <= 0 || n_ops > MI355GS_LOSS_PROGRAM_MAX || !ops || !consts) return MI355GS_EINVAL;
  p.n = n_ops;
  int depth = 0;   // a malformed program (stack underflow, two values left) is refused here, not executed
  for (int i = 0; i < MI355GS_LOSS_PROGRAM_MAX; ++i) {
    p.op[i] = 0; p.k[i] = 0.f;
    if (i >= n_ops) continue;
    const int op = ops[i];
    if (op == MI355GS_LOSS_OP_L1 || op == MI355GS_LOSS_OP_SSIM) ++depth;
    else if (op == MI355GS_LOSS_OP_ADD || op == MI355GS_LOSS_OP_SUB) { if (depth < 2) return MI355GS_EINVAL; --depth; }
    else if (op >= MI355GS_LOSS_OP_MULK && op <= MI355GS_LOSS_OP_NEG) { if (depth < 1) return MI355GS_EINVAL; }
    else return MI355GS_EINVAL;
    p.op[i] = (signed char)op; p.k[i] = consts[i];
  }
  return depth == 1 ? MI355GS_OK : MI355GS_EINVAL;
}

int mi355gs_loss_program_eval(void* stream_, int n_ops, const int32_t* ops, const float* consts, int B, int C, int H, int W,
                              const void* scratch, float* ssim_mean, float* l1_mean, float* out, float* host_out, float ticket) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (!scratch || !l1_mean || !ssim_mean || !out) return MI355GS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ((uintptr_t)host_out & 7) != 0) return MI355GS_EINVAL;
  GsLossProgram p;
  if (build_loss_program(n_ops, ops, consts, p) != MI355GS_OK) return MI355GS_EINVAL;
  GS_KRANGE("loss_program");
  hipLaunchKernelGGL(k_loss_program, dim3(1), dim3(1024), 0, stream, p, fused_nblocks(B, C, H, W), 1.0 / ((double)B * C * H * W),
                     (const float*)scratch, ssim_mean, l1_mean, out, host_out, ticket);
  GS_CHECK_LAUNCH("loss_program");
  return MI355GS_OK;
}

int mi355gs_loss_program_eval_grad(void* stream_, int n_ops, const int32_t* ops, const float* consts, int B, int C, int H, int W,
                                   const void* scratch, float* ssim_mean, float* l1_mean, float* out, float* host_out, float ticket,
                                   const float* img1, const float* img2, const float* dssim_dimg1, float c_l1, float c_ssim, float* d_img1) {
  GS_RANGE();
  hipStream_t stream = (hipStream_t)stream_;
  const int debug = 0;
  if (!scratch || !l1_mean || !ssim_mean || !out || !img1 || !img2 || !d_img1) return MI355GS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ((uintptr_t)host_out & 7) != 0) return MI355GS_EINVAL;
  if (c_ssim != 0.0f && !dssim_dimg1) return MI355GS_EINVAL;
  GsLossProgram p;
  if (build_loss_program(n_ops, ops, consts, p) != MI355GS_OK) return MI355GS_EINVAL;
  const long long n = (long long)B * C * H * W;
  GS_KRANGE("loss_program_grad");
  hipLaunchKernelGGL(k_loss_program_grad, dim3(1u + (unsigned)((n + 4095) / 4096)), dim3(1024), 0, stream, p, fused_nblocks(B, C, H, W),
                     1.0 / (double)n, (const float*)scratch, ssim_mean, l1_mean, out, host_out, ticket, n, img1, img2, dssim_dimg1, c_l1, c_ssim,
                     (float)n, d_img1);
  GS_CHECK_LAUNCH("loss_program_grad");
  return MI355GS_OK;
}

}  // extern "C"

// one-call train step: loss gradient (d loss = 1) in one launch; the loss VALUE is finished later from `scratch`
// (gs_loss_partials_info) by a kernel the step runs anyway
int gs_loss_fused(hipStream_t stream, int C, int H, int W, const float* img1, const float* img2, float lambda_dssim, float* dL_dimg1,
                  void* scratch) {
  const int debug = 0;
  const double inv_n = 1.0 / ((double)C * H * W);
  const dim3 grid((W + FT - 1) / FT, (H + FT - 1) / FT, C);
  {
    GsProfScope prof(3, stream);
    GS_KRANGE("l1_ssim_fused");
    hipLaunchKernelGGL(k_l1_ssim_fused, grid, dim3(FTHREADS), 0, stream, H, W, img1, img2, (float)(-(double)lambda_dssim * inv_n),
                       (float)((1.0 - (double)lambda_dssim) * inv_n), dL_dimg1, (float*)scratch);
  }
  GS_CHECK_LAUNCH("l1_ssim_fused");
  return MI355GS_OK;
}
int gs_loss_fused_nblocks(int C, int H, int W) { return fused_nblocks(1, C, H, W); }
