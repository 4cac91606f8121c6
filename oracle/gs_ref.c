/*
 * ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.
 *
 * Plain-C, tile-based CPU restatement of the differentiable 3D-Gaussian rasterizer that the
 * reference reaches through GaussianRasterizer.forward / autograd backward
 * (reference gaussian_renderer/__init__.py:126-135, train.py:177).  The operator's own source
 * (graphdeco-inria/diff-gaussian-rasterization) is an un-vendored submodule with no pinned revision
 * (reference .gitmodules:4-6), so this file follows the published algorithm as written down in
 * SURVEY.md Appendix A.1-A.5 and is cross-checked against oracle/raster_torch.py (autograd, fp64)
 * in tests/test_oracle.py.  No golden output of the CUDA operator exists: "parity unpinned".
 *
 * dL/dscale under scale_modifier != 1: the covariance is built from s = mod * scale, and the published operator's computeCov3D
 * backward (recalled: `glm::vec3 s = mod * scale` formed first, then `dL_dscale->x = glm::dot(Rt[0], dL_dMt[0])`) returns dL/ds —
 * the gradient with respect to the MODIFIED scale — as dL_dscale.  That is the default here (and in the HIP kernel,
 * csrc/preprocess.hip): "results identical to the reference's" is the bar.  gsref_set_scale_grad_exact(1) switches both
 * precisions to the true derivative, mod x dL/ds (what autograd of oracle/raster_torch.py gives with SCALE_GRAD_EXACT = True);
 * tests/edge_cases.py runs mod = 1.7 in both modes.  At mod = 1 — the only value the reference trains with
 * (gaussian_renderer/__init__.py:29, scaling_modifier=1.0; viewers change it for rendering only) — the two are the same number.
 *
 * Compiled twice (oracle/Makefile): -DGSREF_DOUBLE=0 -> symbols *_f32, =1 -> *_f64.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if GSREF_DOUBLE
typedef double real;
#define SUF(name) name##_f64
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FABS fabs
#else
typedef float real;
#define SUF(name) name##_f32
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FABS fabsf
#endif

#define TILE 16
extern int gsref_scale_grad_exact;   /* defined once, in the f32 object (gsref_set_scale_grad_exact) */

static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
                              (real)0.3731763325901154, (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

typedef struct {
  int P, D, M, W, H, gx, gy;
  real fx, fy, tanfovx, tanfovy, mod;
  real view[16], proj[16], campos[3], bg[3];
  int use_shs, use_cov_precomp;
  /* per Gaussian */
  real *depth, *px, *py, *conic /*3*/, *opac, *rgb /*3*/, *cov3D /*6*/;
  int *radii, *rect /*4*/;
  uint8_t* clamped; /*3*/
  /* binning */
  int64_t R;
  int64_t* tile_start; /* gx*gy+1 */
  int32_t* list;       /* R gaussian ids, per tile, front to back */
  /* per pixel */
  real* final_T;
  int32_t* n_contrib;
} SUF(GsRefCtx);
typedef SUF(GsRefCtx) Ctx;

static inline void tp43(const real* m, const real* p, real* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void tp44(const real* m, const real* p, real* o) {
  tp43(m, p, o);
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void quat_R(const real* q, real R[9]) {
  real r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z); R[2] = 2 * (x * z + r * y);
  R[3] = 2 * (x * y + r * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
  R[6] = 2 * (x * z - r * y); R[7] = 2 * (y * z + r * x); R[8] = 1 - 2 * (x * x + y * y);
}

/* SH basis values b[0..(D+1)^2) for unit direction d (order of reference utils/sh_utils.py:74-100) */
static void sh_basis(int D, const real* d, real* b) {
  real x = d[0], y = d[1], z = d[2];
  b[0] = SH_C0;
  if (D > 0) {
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (D > 1) {
      real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2 * zz - xx - yy);
      b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
      if (D > 2) {
        b[9] = SH_C3[0] * y * (3 * xx - yy); b[10] = SH_C3[1] * xy * z;
        b[11] = SH_C3[2] * y * (4 * zz - xx - yy); b[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
        b[13] = SH_C3[4] * x * (4 * zz - xx - yy); b[14] = SH_C3[5] * z * (xx - yy);
        b[15] = SH_C3[6] * x * (xx - 3 * yy);
      }
    }
  }
}
/* d basis / d(x,y,z): db[k][3] */
static void sh_basis_grad(int D, const real* d, real db[16][3]) {
  real x = d[0], y = d[1], z = d[2];
  memset(db, 0, sizeof(real) * 16 * 3);
  if (D > 0) {
    db[1][1] = -SH_C1; db[2][2] = SH_C1; db[3][0] = -SH_C1;
    if (D > 1) {
      real xx = x * x, yy = y * y, zz = z * z;
      db[4][0] = SH_C2[0] * y; db[4][1] = SH_C2[0] * x;
      db[5][1] = SH_C2[1] * z; db[5][2] = SH_C2[1] * y;
      db[6][0] = SH_C2[2] * -2 * x; db[6][1] = SH_C2[2] * -2 * y; db[6][2] = SH_C2[2] * 4 * z;
      db[7][0] = SH_C2[3] * z; db[7][2] = SH_C2[3] * x;
      db[8][0] = SH_C2[4] * 2 * x; db[8][1] = SH_C2[4] * -2 * y;
      if (D > 2) {
        db[9][0] = SH_C3[0] * 6 * x * y; db[9][1] = SH_C3[0] * (3 * xx - 3 * yy);
        db[10][0] = SH_C3[1] * y * z; db[10][1] = SH_C3[1] * x * z; db[10][2] = SH_C3[1] * x * y;
        db[11][0] = SH_C3[2] * -2 * x * y; db[11][1] = SH_C3[2] * (4 * zz - xx - 3 * yy); db[11][2] = SH_C3[2] * 8 * y * z;
        db[12][0] = SH_C3[3] * -6 * x * z; db[12][1] = SH_C3[3] * -6 * y * z; db[12][2] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
        db[13][0] = SH_C3[4] * (4 * zz - 3 * xx - yy); db[13][1] = SH_C3[4] * -2 * x * y; db[13][2] = SH_C3[4] * 8 * x * z;
        db[14][0] = SH_C3[5] * 2 * x * z; db[14][1] = SH_C3[5] * -2 * y * z; db[14][2] = SH_C3[5] * (xx - yy);
        db[15][0] = SH_C3[6] * (3 * xx - 3 * yy); db[15][1] = SH_C3[6] * -6 * x * y;
      }
    }
  }
}

typedef struct { uint64_t key; int32_t id; } KV;
static int kv_cmp(const void* a, const void* b) {
  const KV* x = (const KV*)a; const KV* y = (const KV*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return (x->id > y->id) - (x->id < y->id);
}

void SUF(gsref_free)(void* c_) {
  Ctx* c = (Ctx*)c_;
  if (!c) return;
  free(c->depth); free(c->px); free(c->py); free(c->conic); free(c->opac); free(c->rgb); free(c->cov3D);
  free(c->radii); free(c->rect); free(c->clamped); free(c->tile_start); free(c->list); free(c->final_T); free(c->n_contrib);
  free(c);
}

int64_t SUF(gsref_num_rendered)(void* c_) { return ((Ctx*)c_)->R; }

/* Forward.  Returns an opaque context (saved state for backward) or NULL. */
void* SUF(gsref_forward)(int P, int D, int M, int W, int H, const real* means3D, const real* shs,
                         const real* colors_precomp, const real* opacities, const real* scales, real mod,
                         const real* rotations, const real* cov3D_precomp, const real* view, const real* proj,
                         const real* campos, real tanfovx, real tanfovy, const real* bg, real* out_color,
                         int32_t* radii_out) {
  Ctx* c = (Ctx*)calloc(1, sizeof(Ctx));
  c->P = P; c->D = D; c->M = M; c->W = W; c->H = H;
  c->gx = (W + TILE - 1) / TILE; c->gy = (H + TILE - 1) / TILE;
  c->tanfovx = tanfovx; c->tanfovy = tanfovy; c->mod = mod;
  c->fx = W / (2 * tanfovx); c->fy = H / (2 * tanfovy);
  memcpy(c->view, view, sizeof(real) * 16); memcpy(c->proj, proj, sizeof(real) * 16);
  memcpy(c->campos, campos, sizeof(real) * 3); memcpy(c->bg, bg, sizeof(real) * 3);
  c->use_shs = colors_precomp == NULL; c->use_cov_precomp = cov3D_precomp != NULL;
  size_t Pn = P > 0 ? (size_t)P : 1;
  c->depth = (real*)calloc(Pn, sizeof(real)); c->px = (real*)calloc(Pn, sizeof(real)); c->py = (real*)calloc(Pn, sizeof(real));
  c->conic = (real*)calloc(Pn * 3, sizeof(real)); c->opac = (real*)calloc(Pn, sizeof(real)); c->rgb = (real*)calloc(Pn * 3, sizeof(real));
  c->cov3D = (real*)calloc(Pn * 6, sizeof(real)); c->radii = (int*)calloc(Pn, sizeof(int)); c->rect = (int*)calloc(Pn * 4, sizeof(int));
  c->clamped = (uint8_t*)calloc(Pn * 3, 1);
  const int T = c->gx * c->gy;
  c->tile_start = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));
  c->final_T = (real*)calloc((size_t)W * H, sizeof(real));
  c->n_contrib = (int32_t*)calloc((size_t)W * H, sizeof(int32_t));
  int* touched = (int*)calloc(Pn, sizeof(int));

  /* ---- A.1 preprocess */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) {
    radii_out[i] = 0; touched[i] = 0;
    const real* m = means3D + 3 * (size_t)i;
    real pv[3]; tp43(view, m, pv);
    if (pv[2] <= (real)0.2) continue;
    real ph[4]; tp44(proj, m, ph);
    real pw = 1 / (ph[3] + (real)0.0000001);
    real ndcx = ph[0] * pw, ndcy = ph[1] * pw;
    real* cov = c->cov3D + 6 * (size_t)i;
    if (cov3D_precomp) memcpy(cov, cov3D_precomp + 6 * (size_t)i, sizeof(real) * 6);
    else {
      real Rm[9]; quat_R(rotations + 4 * (size_t)i, Rm);
      const real* s = scales + 3 * (size_t)i;
      real L[9];
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) L[r * 3 + k] = Rm[r * 3 + k] * (mod * s[k]);
      real S[9];
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) {
        real acc = 0; for (int j = 0; j < 3; ++j) acc += L[r * 3 + j] * L[k * 3 + j];
        S[r * 3 + k] = acc;
      }
      cov[0] = S[0]; cov[1] = S[1]; cov[2] = S[2]; cov[3] = S[4]; cov[4] = S[5]; cov[5] = S[8];
    }
    /* EWA */
    real limx = (real)1.3 * tanfovx, limy = (real)1.3 * tanfovy;
    real txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    real tx = fmin(limx, fmax(-limx, txtz)) * pv[2];
    real ty = fmin(limy, fmax(-limy, tytz)) * pv[2];
    real tz = pv[2];
    real J[6] = {c->fx / tz, 0, -(c->fx * tx) / (tz * tz), 0, c->fy / tz, -(c->fy * ty) / (tz * tz)};
    real Wm[9] = {view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]};
    real Mx[6];
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) Mx[r * 3 + k] = J[r * 3] * Wm[k] + J[r * 3 + 1] * Wm[3 + k] + J[r * 3 + 2] * Wm[6 + k];
    real Sg[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
    real MS[6];
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) MS[r * 3 + k] = Mx[r * 3] * Sg[k] + Mx[r * 3 + 1] * Sg[3 + k] + Mx[r * 3 + 2] * Sg[6 + k];
    real a = MS[0] * Mx[0] + MS[1] * Mx[1] + MS[2] * Mx[2] + (real)0.3;
    real b = MS[0] * Mx[3] + MS[1] * Mx[4] + MS[2] * Mx[5];
    real cc = MS[3] * Mx[3] + MS[4] * Mx[4] + MS[5] * Mx[5] + (real)0.3;
    real det = a * cc - b * b;
    if (det == 0) continue;
    real dinv = 1 / det;
    real mid = (real)0.5 * (a + cc);
    real lam1 = mid + R_SQRT(fmax((real)0.1, mid * mid - det));
    real lam2 = mid - R_SQRT(fmax((real)0.1, mid * mid - det));
    real radius = R_CEIL(3 * R_SQRT(fmax(lam1, lam2)));
    real px = ((ndcx + 1) * W - 1) * (real)0.5, py = ((ndcy + 1) * H - 1) * (real)0.5;
    int rminx = clampi((int)((px - radius) / TILE), 0, c->gx), rminy = clampi((int)((py - radius) / TILE), 0, c->gy);
    int rmaxx = clampi((int)((px + radius + TILE - 1) / TILE), 0, c->gx), rmaxy = clampi((int)((py + radius + TILE - 1) / TILE), 0, c->gy);
    if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
    real* rgb = c->rgb + 3 * (size_t)i;
    if (colors_precomp) { rgb[0] = colors_precomp[3 * (size_t)i]; rgb[1] = colors_precomp[3 * (size_t)i + 1]; rgb[2] = colors_precomp[3 * (size_t)i + 2]; }
    else {
      real d[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
      real n = R_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      d[0] /= n; d[1] /= n; d[2] /= n;
      real bs[16]; sh_basis(D, d, bs);
      const real* sh = shs + (size_t)i * M * 3;
      int nb = (D + 1) * (D + 1);
      for (int ch = 0; ch < 3; ++ch) {
        real acc = 0; for (int k = 0; k < nb; ++k) acc += bs[k] * sh[k * 3 + ch];
        acc += (real)0.5;
        c->clamped[3 * (size_t)i + ch] = acc < 0;
        rgb[ch] = acc < 0 ? 0 : acc;
      }
    }
    c->depth[i] = pv[2]; c->radii[i] = (int)radius; radii_out[i] = (int)radius;
    c->px[i] = px; c->py[i] = py;
    c->conic[3 * (size_t)i] = cc * dinv; c->conic[3 * (size_t)i + 1] = -b * dinv; c->conic[3 * (size_t)i + 2] = a * dinv;
    c->opac[i] = opacities[i];
    c->rect[4 * (size_t)i] = rminx; c->rect[4 * (size_t)i + 1] = rminy; c->rect[4 * (size_t)i + 2] = rmaxx; c->rect[4 * (size_t)i + 3] = rmaxy;
    touched[i] = (rmaxx - rminx) * (rmaxy - rminy);
  }

  /* ---- A.2 binning: per-tile lists sorted by (depth bits, gaussian index) */
  int64_t R = 0;
  for (int i = 0; i < P; ++i) R += touched[i];
  c->R = R;
  KV* kv = (KV*)malloc(sizeof(KV) * (size_t)(R > 0 ? R : 1));
  {
    int64_t off = 0;
    for (int i = 0; i < P; ++i) {
      if (!touched[i]) continue;
      float df = (float)c->depth[i]; uint32_t dbits; memcpy(&dbits, &df, 4);
      const int* rc = c->rect + 4 * (size_t)i;
      for (int y = rc[1]; y < rc[3]; ++y) for (int x = rc[0]; x < rc[2]; ++x) {
        kv[off].key = ((uint64_t)(y * c->gx + x) << 32) | dbits; kv[off].id = i; ++off;
      }
    }
  }
  qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
  c->list = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
  {
    int64_t j = 0;
    for (int t = 0; t < T; ++t) {
      c->tile_start[t] = j;
      while (j < R && (int)(kv[j].key >> 32) == t) { c->list[j] = kv[j].id; ++j; }
    }
    c->tile_start[T] = R;
  }
  free(kv); free(touched);

  /* ---- A.3 composite */
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < T; ++t) {
    int tx0 = (t % c->gx) * TILE, ty0 = (t / c->gx) * TILE;
    int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
    for (int py_ = ty0; py_ < ty0 + TILE && py_ < H; ++py_) for (int px_ = tx0; px_ < tx0 + TILE && px_ < W; ++px_) {
      real Tr = 1, C[3] = {0, 0, 0}; int contributor = 0, last = 0;
      for (int64_t j = s; j < e; ++j) {
        int g = c->list[j]; ++contributor;
        real dx = c->px[g] - (real)px_, dy = c->py[g] - (real)py_;
        const real* co = c->conic + 3 * (size_t)g;
        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0) continue;
        real alpha = fmin((real)0.99, c->opac[g] * R_EXP(power));
        if (alpha < (real)(1.0 / 255.0)) continue;
        real testT = Tr * (1 - alpha);
        if (testT < (real)0.0001) break;
        const real* col = c->rgb + 3 * (size_t)g;
        for (int ch = 0; ch < 3; ++ch) C[ch] += col[ch] * alpha * Tr;
        Tr = testT; last = contributor;
      }
      size_t pix = (size_t)py_ * W + px_;
      c->final_T[pix] = Tr; c->n_contrib[pix] = last;
      for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix] = C[ch] + Tr * bg[ch];
    }
  }
  return c;
}

/* Backward (A.4/A.5).  Every output must hold P*k reals; all are overwritten. */
void SUF(gsref_backward)(void* c_, const real* means3D, const real* shs, const real* scales, const real* rotations,
                         const real* dL_dpix, real* dL_dmeans3D, real* dL_dmeans2D, real* dL_dshs, real* dL_dcolors,
                         real* dL_dopac, real* dL_dscales, real* dL_drots, real* dL_dcov3D) {
  Ctx* c = (Ctx*)c_;
  const int P = c->P, W = c->W, H = c->H, T = c->gx * c->gy;
  const size_t Pn = P > 0 ? (size_t)P : 1;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
  if (nthreads > 16) nthreads = 16; /* per-thread accumulators: bound memory (16 x P x 9 reals) and the fixed-order reduction */
#endif
  /* per-thread accumulators: [mean2D.x, mean2D.y, conic a, b(full), c, opacity, r, g, b] */
  real* acc = (real*)calloc((size_t)nthreads * Pn * 9, sizeof(real));
#pragma omp parallel num_threads(nthreads)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    real* A = acc + (size_t)tid * Pn * 9;
#pragma omp for schedule(static)
    for (int t = 0; t < T; ++t) {
      int tx0 = (t % c->gx) * TILE, ty0 = (t / c->gx) * TILE;
      int64_t s = c->tile_start[t];
      for (int py_ = ty0; py_ < ty0 + TILE && py_ < H; ++py_) for (int px_ = tx0; px_ < tx0 + TILE && px_ < W; ++px_) {
        size_t pix = (size_t)py_ * W + px_;
        const real Tfinal = c->final_T[pix];
        real Tr = Tfinal; int last = c->n_contrib[pix];
        real gpix[3] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[2 * (size_t)H * W + pix]};
        real accum_rec[3] = {0, 0, 0}, last_alpha = 0, last_color[3] = {0, 0, 0};
        real bg_dot = c->bg[0] * gpix[0] + c->bg[1] * gpix[1] + c->bg[2] * gpix[2];
        for (int k = last; k >= 1; --k) {
          int g = c->list[s + k - 1];
          real dx = c->px[g] - (real)px_, dy = c->py[g] - (real)py_;
          const real* co = c->conic + 3 * (size_t)g;
          real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0) continue;
          real G = R_EXP(power);
          real alpha = fmin((real)0.99, c->opac[g] * G);
          if (alpha < (real)(1.0 / 255.0)) continue;
          Tr = Tr / (1 - alpha);
          real dchannel = alpha * Tr;
          real dL_dalpha = 0;
          const real* col = c->rgb + 3 * (size_t)g;
          real* Ag = A + 9 * (size_t)g;
          for (int ch = 0; ch < 3; ++ch) {
            accum_rec[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum_rec[ch];
            last_color[ch] = col[ch];
            dL_dalpha += (col[ch] - accum_rec[ch]) * gpix[ch];
            Ag[6 + ch] += dchannel * gpix[ch];
          }
          dL_dalpha *= Tr;
          last_alpha = alpha;
          dL_dalpha += (-Tfinal / (1 - alpha)) * bg_dot;
          real dL_dG = c->opac[g] * dL_dalpha;
          real gdx = G * dx, gdy = G * dy;
          real dG_ddelx = -gdx * co[0] - gdy * co[1];
          real dG_ddely = -gdy * co[2] - gdx * co[1];
          Ag[0] += dL_dG * dG_ddelx * (real)0.5 * W;
          Ag[1] += dL_dG * dG_ddely * (real)0.5 * H;
          Ag[2] += (real)-0.5 * gdx * dx * dL_dG;
          Ag[3] += -gdx * dy * dL_dG; /* full d/db of the packed off-diagonal */
          Ag[4] += (real)-0.5 * gdy * dy * dL_dG;
          Ag[5] += G * dL_dalpha;
        }
      }
    }
  }
  /* fixed-order reduction over threads */
  for (int th = 1; th < nthreads; ++th) {
    real* A = acc + (size_t)th * Pn * 9;
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < Pn * 9; ++k) acc[k] += A[k];
  }

  /* ---- per-Gaussian chain (A.5) */
  const int M = c->M, D = c->D;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) {
    const real* Ag = acc + 9 * (size_t)i;
    real* gm = dL_dmeans3D + 3 * (size_t)i; gm[0] = gm[1] = gm[2] = 0;
    real* g2 = dL_dmeans2D + 3 * (size_t)i; g2[0] = Ag[0]; g2[1] = Ag[1]; g2[2] = 0;
    dL_dopac[i] = Ag[5];
    if (dL_dcolors) { dL_dcolors[3 * (size_t)i] = Ag[6]; dL_dcolors[3 * (size_t)i + 1] = Ag[7]; dL_dcolors[3 * (size_t)i + 2] = Ag[8]; }
    if (dL_dshs) memset(dL_dshs + (size_t)i * M * 3, 0, sizeof(real) * M * 3);
    real gcov[6] = {0, 0, 0, 0, 0, 0};
    if (dL_dscales) { real* p = dL_dscales + 3 * (size_t)i; p[0] = p[1] = p[2] = 0; }
    if (dL_drots) { real* p = dL_drots + 4 * (size_t)i; p[0] = p[1] = p[2] = p[3] = 0; }
    if (dL_dcov3D) memset(dL_dcov3D + 6 * (size_t)i, 0, sizeof(real) * 6);
    if (c->radii[i] <= 0) continue;
    const real* m = means3D + 3 * (size_t)i;
    const real* view = c->view; const real* proj = c->proj;
    /* conic -> cov2D -> cov3D, mean (computeCov2D backward) */
    {
      real pv[3]; tp43(view, m, pv);
      real limx = (real)1.3 * c->tanfovx, limy = (real)1.3 * c->tanfovy;
      real txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
      real tx = fmin(limx, fmax(-limx, txtz)) * pv[2], ty = fmin(limy, fmax(-limy, tytz)) * pv[2], tz = pv[2];
      real xmul = (txtz < -limx || txtz > limx) ? 0 : 1, ymul = (tytz < -limy || tytz > limy) ? 0 : 1;
      real J[6] = {c->fx / tz, 0, -(c->fx * tx) / (tz * tz), 0, c->fy / tz, -(c->fy * ty) / (tz * tz)};
      real Wm[9] = {view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]};
      real Mx[6];
      for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) Mx[r * 3 + k] = J[r * 3] * Wm[k] + J[r * 3 + 1] * Wm[3 + k] + J[r * 3 + 2] * Wm[6 + k];
      const real* cov = c->cov3D + 6 * (size_t)i;
      real Sg[9] = {cov[0], cov[1], cov[2], cov[1], cov[3], cov[4], cov[2], cov[4], cov[5]};
      real MS[6];
      for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) MS[r * 3 + k] = Mx[r * 3] * Sg[k] + Mx[r * 3 + 1] * Sg[3 + k] + Mx[r * 3 + 2] * Sg[6 + k];
      real a = MS[0] * Mx[0] + MS[1] * Mx[1] + MS[2] * Mx[2] + (real)0.3;
      real b = MS[0] * Mx[3] + MS[1] * Mx[4] + MS[2] * Mx[5];
      real cc = MS[3] * Mx[3] + MS[4] * Mx[4] + MS[5] * Mx[5] + (real)0.3;
      real det = a * cc - b * b;
      real Dv = 1 / (det * det + (real)0.0000001);
      real gA = Ag[2], gB = Ag[3], gC = Ag[4];
      real ga = Dv * (-cc * cc * gA + b * cc * gB + (det - a * cc) * gC);
      real gc = Dv * ((det - a * cc) * gA + a * b * gB - a * a * gC);
      real gb = Dv * (2 * b * cc * gA - (det + 2 * b * b) * gB + 2 * a * b * gC);
      /* gSigma_full = M^T Gm M, Gm = [[ga, gb/2],[gb/2, gc]] ; packed off-diagonals carry both entries */
      real Gm[4] = {ga, (real)0.5 * gb, (real)0.5 * gb, gc};
      real GM[6];
      for (int k = 0; k < 3; ++k) { GM[k] = Gm[0] * Mx[k] + Gm[1] * Mx[3 + k]; GM[3 + k] = Gm[2] * Mx[k] + Gm[3] * Mx[3 + k]; }
      real F[9];
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) F[r * 3 + k] = Mx[r] * GM[k] + Mx[3 + r] * GM[3 + k];
      gcov[0] = F[0]; gcov[1] = 2 * F[1]; gcov[2] = 2 * F[2]; gcov[3] = F[4]; gcov[4] = 2 * F[5]; gcov[5] = F[8];
      /* gM = 2 Gm M Sigma ; gJ = gM W^T */
      real gM[6];
      for (int k = 0; k < 3; ++k) {
        gM[k] = 2 * (GM[0] * Sg[k] + GM[1] * Sg[3 + k] + GM[2] * Sg[6 + k]);
        gM[3 + k] = 2 * (GM[3] * Sg[k] + GM[4] * Sg[3 + k] + GM[5] * Sg[6 + k]);
      }
      real gJ[6];
      for (int r = 0; r < 2; ++r) for (int k = 0; k < 3; ++k) gJ[r * 3 + k] = gM[r * 3] * Wm[k * 3] + gM[r * 3 + 1] * Wm[k * 3 + 1] + gM[r * 3 + 2] * Wm[k * 3 + 2];
      real tz2 = 1 / (tz * tz), tz3 = tz2 / tz;
      real gtx = xmul * -c->fx * tz2 * gJ[2];
      real gty = ymul * -c->fy * tz2 * gJ[5];
      real gtz = -c->fx * tz2 * gJ[0] - c->fy * tz2 * gJ[4] + 2 * c->fx * tx * tz3 * gJ[2] + 2 * c->fy * ty * tz3 * gJ[5];
      /* g mean = (rotation part of view)^T g t : transformVec4x3Transpose */
      gm[0] = view[0] * gtx + view[1] * gty + view[2] * gtz;
      gm[1] = view[4] * gtx + view[5] * gty + view[6] * gtz;
      gm[2] = view[8] * gtx + view[9] * gty + view[10] * gtz;
    }
    /* projection path */
    {
      real ph[4]; tp44(proj, m, ph);
      real mw = 1 / (ph[3] + (real)0.0000001);
      real mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
      gm[0] += (proj[0] * mw - proj[3] * mul1) * Ag[0] + (proj[1] * mw - proj[3] * mul2) * Ag[1];
      gm[1] += (proj[4] * mw - proj[7] * mul1) * Ag[0] + (proj[5] * mw - proj[7] * mul2) * Ag[1];
      gm[2] += (proj[8] * mw - proj[11] * mul1) * Ag[0] + (proj[9] * mw - proj[11] * mul2) * Ag[1];
    }
    /* SH */
    if (c->use_shs) {
      real d0[3] = {m[0] - c->campos[0], m[1] - c->campos[1], m[2] - c->campos[2]};
      real n = R_SQRT(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
      real d[3] = {d0[0] / n, d0[1] / n, d0[2] / n};
      real bs[16]; sh_basis(D, d, bs);
      real db[16][3]; sh_basis_grad(D, d, db);
      int nb = (D + 1) * (D + 1);
      const real* sh = shs + (size_t)i * M * 3;
      real* gsh = dL_dshs + (size_t)i * M * 3;
      real gd[3] = {0, 0, 0};
      for (int ch = 0; ch < 3; ++ch) {
        real gr = c->clamped[3 * (size_t)i + ch] ? 0 : Ag[6 + ch];
        for (int k = 0; k < nb; ++k) {
          gsh[k * 3 + ch] = bs[k] * gr;
          gd[0] += db[k][0] * sh[k * 3 + ch] * gr; gd[1] += db[k][1] * sh[k * 3 + ch] * gr; gd[2] += db[k][2] * sh[k * 3 + ch] * gr;
        }
      }
      /* through d = v/|v| */
      real dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
      gm[0] += (gd[0] - d[0] * dot) / n; gm[1] += (gd[1] - d[1] * dot) / n; gm[2] += (gd[2] - d[2] * dot) / n;
    }
    /* cov3D -> scale, rotation */
    if (c->use_cov_precomp) { if (dL_dcov3D) memcpy(dL_dcov3D + 6 * (size_t)i, gcov, sizeof(real) * 6); }
    else {
      const real* q = rotations + 4 * (size_t)i; const real* s = scales + 3 * (size_t)i;
      real Rm[9]; quat_R(q, Rm);
      real L[9]; for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) L[r * 3 + k] = Rm[r * 3 + k] * (c->mod * s[k]);
      real Gs[9] = {gcov[0], (real)0.5 * gcov[1], (real)0.5 * gcov[2], (real)0.5 * gcov[1], gcov[3], (real)0.5 * gcov[4], (real)0.5 * gcov[2], (real)0.5 * gcov[4], gcov[5]};
      real gL[9];
      for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) gL[r * 3 + k] = 2 * (Gs[r * 3] * L[k] + Gs[r * 3 + 1] * L[3 + k] + Gs[r * 3 + 2] * L[6 + k]);
      real* gs = dL_dscales + 3 * (size_t)i;
      real Rp[9];
      for (int k = 0; k < 3; ++k) {
        gs[k] = (gsref_scale_grad_exact ? c->mod : (real)1) * (Rm[k] * gL[k] + Rm[3 + k] * gL[3 + k] + Rm[6 + k] * gL[6 + k]);
        for (int r = 0; r < 3; ++r) Rp[r * 3 + k] = gL[r * 3 + k] * c->mod * s[k];
      }
      real r_ = q[0], x = q[1], y = q[2], z = q[3];
      real* gq = dL_drots + 4 * (size_t)i;
#define RP(i_, j_) Rp[(i_) * 3 + (j_)]
      gq[0] = 2 * (z * (RP(1, 0) - RP(0, 1)) + y * (RP(0, 2) - RP(2, 0)) + x * (RP(2, 1) - RP(1, 2)));
      gq[1] = 2 * (y * (RP(0, 1) + RP(1, 0)) + z * (RP(0, 2) + RP(2, 0)) + r_ * (RP(2, 1) - RP(1, 2))) - 4 * x * (RP(1, 1) + RP(2, 2));
      gq[2] = 2 * (x * (RP(0, 1) + RP(1, 0)) + r_ * (RP(0, 2) - RP(2, 0)) + z * (RP(1, 2) + RP(2, 1))) - 4 * y * (RP(0, 0) + RP(2, 2));
      gq[3] = 2 * (r_ * (RP(1, 0) - RP(0, 1)) + x * (RP(0, 2) + RP(2, 0)) + y * (RP(1, 2) + RP(2, 1))) - 4 * z * (RP(0, 0) + RP(1, 1));
#undef RP
    }
  }
  free(acc);
}

#if !GSREF_DOUBLE
/* dL/dscale convention of both precisions (see the header): 0 = dL/d(mod * scale) as the published operator, 1 = true derivative */
int gsref_scale_grad_exact = 0;
int gsref_set_scale_grad_exact(int on) { const int old = gsref_scale_grad_exact; if (on >= 0) gsref_scale_grad_exact = on ? 1 : 0; return old; }
/* thread count used by both precisions (bench.py's cpu_baseline reports it as `cores`) */
int gsref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}
#endif

/* introspection for tests */
void SUF(gsref_get_aux)(void* c_, real* final_T, int32_t* n_contrib, int64_t* tile_start, int32_t* list) {
  Ctx* c = (Ctx*)c_;
  if (final_T) memcpy(final_T, c->final_T, sizeof(real) * (size_t)c->W * c->H);
  if (n_contrib) memcpy(n_contrib, c->n_contrib, sizeof(int32_t) * (size_t)c->W * c->H);
  if (tile_start) memcpy(tile_start, c->tile_start, sizeof(int64_t) * ((size_t)c->gx * c->gy + 1));
  if (list) memcpy(list, c->list, sizeof(int32_t) * (size_t)c->R);
}
