// TEST INFRASTRUCTURE ONLY — a minimal SIMT emulator that stands in for <hip/hip_runtime.h>
// when the kernel sources under instantsplat_amd/csrc/ are compiled with plain g++
// (tests/emu/build_emu.sh puts this directory first on the include path).
//
// Why it exists: the build container has no GPU, and GPU time is scarce.  Compiling the
// *unmodified* .hip sources against this header lets the CPU test-suite execute the real
// kernel logic (block barriers, wave64 ballots / shuffles, atomics, LDS staging) against the
// oracle before anything is sent to an MI355X.  It is never loaded by the product path:
// instantsplat_amd/_lib.py only ever opens the hipcc-built libmi355gs.so and raises if it is
// missing.  Only tests/ load the emulated library.
//
// Model: one OS thread; each GPU thread of a block is a ucontext fiber; a block runs to
// completion before the next starts.  Fibers yield at __syncthreads() and at wave-level
// operations; wave ops complete when every live lane of the 64-lane wave has arrived
// (finished lanes count as inactive, as on hardware).  Divergent use of wave ops (lanes of
// one wave waiting at different kinds of sync points) aborts with a diagnostic — the kernels
// are written to call wave ops convergently.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

// ---------------------------------------------------------------- vector types
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

// ---------------------------------------------------------------- runtime API subset
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
// (counted: tests assert that a training iteration of the drop-in path issues no memset of its own — the frame's accumulators are
// cleared by the projection kernel on its way)
inline long g_emu_memset_calls = 0;
extern "C" __attribute__((weak, visibility("default"))) long mi355gs_emu_memset_calls() { return g_emu_memset_calls; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { ++g_emu_memset_calls; memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 20; return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }

// ---------------------------------------------------------------- emulator core
namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 128 * 1024;

enum State { READY, AT_BLOCK_BARRIER, AT_WAVE_OP, DONE };

struct Fiber {
  ucontext_t ctx;
  State state = DONE;
  dim3 tid;
  unsigned flat = 0;
  unsigned long long wave_op_seq = 0;  // number of wave ops this lane has entered
};

struct WaveScratch {
  // double-buffered by op parity so a fast lane entering op n+1 cannot clobber op n's values
  uint64_t val[2][WAVE];
  bool pred[2][WAVE];
  uint64_t active[2];  // lanes that took part in the op held in each buffer (set when it resolves)
};

struct Ctx {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  std::vector<char*> stacks;
  std::vector<WaveScratch> waves;
  Fiber* cur = nullptr;
  dim3 blockIdx, blockDim, gridDim;
  std::function<void()> body;
  unsigned nthreads = 0;
};

inline Ctx& ctx() { static Ctx c; return c; }

inline void fiber_entry() {
  Ctx& c = ctx();
  c.body();
  c.cur->state = DONE;
  swapcontext(&c.cur->ctx, &c.sched);
}

inline void yield_to_sched(State s) {
  Ctx& c = ctx();
  Fiber* f = c.cur;
  f->state = s;
  swapcontext(&f->ctx, &c.sched);
}

inline void run_block() {
  Ctx& c = ctx();
  const unsigned n = c.nthreads;
  const unsigned nwaves = (n + WAVE - 1) / WAVE;
  if (c.fibers.size() < n) c.fibers.resize(n);
  while (c.stacks.size() < n) c.stacks.push_back((char*)malloc(STACK_BYTES));
  if (c.waves.size() < nwaves) c.waves.resize(nwaves);
  for (unsigned t = 0; t < n; ++t) {
    Fiber& f = c.fibers[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = c.stacks[t];
    f.ctx.uc_stack.ss_size = STACK_BYTES;
    f.ctx.uc_link = &c.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    f.state = READY;
    f.flat = t;
    f.tid = dim3(t % c.blockDim.x, (t / c.blockDim.x) % c.blockDim.y, t / (c.blockDim.x * c.blockDim.y));
    f.wave_op_seq = 0;
  }
  for (;;) {
    bool all_done = true;
    // advance every wave until each of its lanes is DONE or parked at the block barrier
    for (unsigned w = 0; w < nwaves; ++w) {
      const unsigned lo = w * WAVE, hi = std::min(n, lo + WAVE);
      for (;;) {
        bool progressed = false;
        for (unsigned t = lo; t < hi; ++t) {
          Fiber& f = c.fibers[t];
          if (f.state == READY) {
            c.cur = &f;
            swapcontext(&c.sched, &f.ctx);
            progressed = true;
          }
        }
        // resolve a pending wave op: every live lane must be waiting on the same op number
        unsigned waiting = 0, barrier = 0, live = 0;
        unsigned long long seq = 0; bool seq_set = false, seq_mismatch = false;
        for (unsigned t = lo; t < hi; ++t) {
          Fiber& f = c.fibers[t];
          if (f.state == DONE) continue;
          ++live;
          if (f.state == AT_WAVE_OP) {
            ++waiting;
            if (!seq_set) { seq = f.wave_op_seq; seq_set = true; }
            else if (seq != f.wave_op_seq) seq_mismatch = true;
          } else if (f.state == AT_BLOCK_BARRIER) ++barrier;
        }
        if (waiting && waiting == live && !seq_mismatch) {
          uint64_t mask = 0;
          for (unsigned t = lo; t < hi; ++t)
            if (c.fibers[t].state == AT_WAVE_OP) { c.fibers[t].state = READY; mask |= 1ull << (t - lo); }
          c.waves[w].active[(seq - 1) & 1] = mask;
          continue;
        }
        if (waiting && (barrier || seq_mismatch) && !progressed) {
          fprintf(stderr, "[hip_emu] divergent wave op in block (%u,%u,%u) wave %u: %u lanes at wave op, %u at barrier, seq mismatch=%d\n",
                  c.blockIdx.x, c.blockIdx.y, c.blockIdx.z, w, waiting, barrier, (int)seq_mismatch);
          abort();
        }
        if (!progressed) break;
      }
    }
    unsigned at_barrier = 0, live = 0;
    for (unsigned t = 0; t < n; ++t) {
      State s = c.fibers[t].state;
      if (s != DONE) { ++live; all_done = false; }
      if (s == AT_BLOCK_BARRIER) ++at_barrier;
    }
    if (all_done) break;
    if (at_barrier == live) {
      for (unsigned t = 0; t < n; ++t)
        if (c.fibers[t].state == AT_BLOCK_BARRIER) c.fibers[t].state = READY;
    } else {
      fprintf(stderr, "[hip_emu] deadlock in block (%u,%u,%u): %u live, %u at barrier\n", c.blockIdx.x, c.blockIdx.y, c.blockIdx.z, live, at_barrier);
      abort();
    }
  }
}

inline std::vector<char>& dyn_shared_buf() { static std::vector<char> b; return b; }
inline void* dyn_shared() { return dyn_shared_buf().data(); }

template <class F>
inline void launch(F&& body, dim3 grid, dim3 block, size_t shmem = 0) {
  Ctx& c = ctx();
  if (dyn_shared_buf().size() < shmem + 16) dyn_shared_buf().resize(shmem + 16);
  c.gridDim = grid;
  c.blockDim = block;
  c.nthreads = block.x * block.y * block.z;
  c.body = body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        c.blockIdx = dim3(bx, by, bz);
        run_block();
      }
}

// ---- wave-op plumbing
inline unsigned lane_id() { return ctx().cur->flat % WAVE; }
inline unsigned wave_id() { return ctx().cur->flat / WAVE; }

// deposit (val,pred) for this lane, wait for the wave, return the buffer index to read from
inline int wave_exchange(uint64_t v, bool p) {
  Ctx& c = ctx();
  Fiber* f = c.cur;
  const int buf = (int)(f->wave_op_seq & 1);
  WaveScratch& ws = c.waves[wave_id()];
  ws.val[buf][lane_id()] = v;
  ws.pred[buf][lane_id()] = p;
  f->wave_op_seq++;
  yield_to_sched(AT_WAVE_OP);
  return buf;
}
inline bool lane_in_op(int buf, unsigned lane) { return (ctx().waves[wave_id()].active[buf] >> lane) & 1; }
template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

}  // namespace emu

#define threadIdx (emu::ctx().cur->tid)
#define blockIdx (emu::ctx().blockIdx)
#define blockDim (emu::ctx().blockDim)
#define gridDim (emu::ctx().gridDim)
#define warpSize 64

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t /*stream*/, A... args) {
  emu::launch([=]() { kernel(args...); }, grid, block, shmem);
}
// HIP's portable spelling of `extern __shared__ type var[]`
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_shared();

// ---------------------------------------------------------------- device intrinsics
static inline void __syncthreads() { emu::yield_to_sched(emu::AT_BLOCK_BARRIER); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

static inline unsigned long long __ballot(int pred) {
  int buf = emu::wave_exchange(0, pred != 0);
  auto& ws = emu::ctx().waves[emu::wave_id()];
  unsigned long long m = 0;
  for (unsigned l = 0; l < 64; ++l)
    if (emu::lane_in_op(buf, l) && ws.pred[buf][l]) m |= 1ull << l;
  return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
  int buf = emu::wave_exchange(0, pred != 0);
  auto& ws = emu::ctx().waves[emu::wave_id()];
  for (unsigned l = 0; l < 64; ++l)
    if (emu::lane_in_op(buf, l) && !ws.pred[buf][l]) return 0;
  return 1;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  unsigned me = emu::lane_id();
  unsigned base = me - (me % width);
  unsigned s = base + ((unsigned)src % (unsigned)width);
  return emu::from_bits<T>(emu::ctx().waves[emu::wave_id()].val[buf][s]);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  unsigned me = emu::lane_id();
  unsigned s = me ^ (unsigned)mask;
  if (s / width != me / width) s = me;
  return emu::from_bits<T>(emu::ctx().waves[emu::wave_id()].val[buf][s]);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  unsigned me = emu::lane_id();
  unsigned s = me + delta;
  if (s / width != me / width) s = me;
  return emu::from_bits<T>(emu::ctx().waves[emu::wave_id()].val[buf][s]);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  unsigned me = emu::lane_id();
  unsigned s = (me % width) >= delta ? me - delta : me;
  return emu::from_bits<T>(emu::ctx().waves[emu::wave_id()].val[buf][s]);
}
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  for (unsigned l = 0; l < 64; ++l)
    if (emu::lane_in_op(buf, l)) return emu::from_bits<unsigned>(emu::ctx().waves[emu::wave_id()].val[buf][l]);
  return v;
}
static inline unsigned __builtin_amdgcn_readlane(unsigned v, int lane) {
  int buf = emu::wave_exchange(emu::to_bits(v), true);
  return emu::from_bits<unsigned>(emu::ctx().waves[emu::wave_id()].val[buf][lane & 63]);
}
// lane index inside the wave (what v_mbcnt_lo/hi(~0, 0) computes on hardware)
static inline unsigned __lane_id() { return emu::lane_id(); }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
  unsigned l = emu::lane_id();
  unsigned lt = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
  return add + __builtin_popcount(mask & lt);
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
  unsigned l = emu::lane_id();
  unsigned lt = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
  return add + __builtin_popcount(mask & lt);
}
// v_mov_b32_dpp semantics (GFX9 DPP): returns src from the lane selected by dpp_ctrl; lanes whose
// source is out of range, or whose row/bank is masked off, keep `old` (bound_ctrl: 0 instead).
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  int buf = emu::wave_exchange(emu::to_bits(src), true);
  const int l = (int)emu::lane_id();
  const int row = l >> 4, r = l & 15;
  int sl = -1;  // source lane
  if (ctrl >= 0x000 && ctrl <= 0x0FF) { sl = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3); }
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { int n = ctrl & 15; sl = (r + n <= 15) ? l + n : -1; }      // row_shl
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { int n = ctrl & 15; sl = (r - n >= 0) ? l - n : -1; }       // row_shr
  else if (ctrl >= 0x121 && ctrl <= 0x12F) { int n = ctrl & 15; sl = (row << 4) | ((r - n) & 15); }     // row_ror
  else if (ctrl == 0x130) { sl = l + 1 <= 63 ? l + 1 : -1; }                                             // wave_shl:1
  else if (ctrl == 0x134) { sl = (l + 1) & 63; }                                                         // wave_rol:1
  else if (ctrl == 0x138) { sl = l - 1 >= 0 ? l - 1 : -1; }                                              // wave_shr:1
  else if (ctrl == 0x13C) { sl = (l - 1) & 63; }                                                         // wave_ror:1
  else if (ctrl == 0x140) { sl = (row << 4) | (15 - r); }                                                // row_mirror
  else if (ctrl == 0x141) { sl = (l & ~7) | (7 - (l & 7)); }                                             // row_half_mirror
  else if (ctrl == 0x142) { sl = row >= 1 ? ((row - 1) << 4) | 15 : -1; }                                // row_bcast:15
  else if (ctrl == 0x143) { sl = row >= 2 ? 31 : -1; }                                                   // row_bcast:31
  else { fprintf(stderr, "[hip_emu] unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (r >> 2)) & 1);
  if (!enabled) return old;
  if (sl < 0 || !emu::lane_in_op(buf, (unsigned)sl)) return bound_ctrl ? 0 : old;
  return emu::from_bits<int>(emu::ctx().waves[emu::wave_id()].val[buf][sl]);
}
// ds_bpermute_b32: every lane reads `src` of the lane whose index is addr / 4 (mod 64)
static inline int __builtin_amdgcn_ds_bpermute(int addr, int src) {
  int buf = emu::wave_exchange(emu::to_bits(src), true);
  const unsigned sl = ((unsigned)addr >> 2) & 63u;
  if (!emu::lane_in_op(buf, sl)) return 0;
  return emu::from_bits<int>(emu::ctx().waves[emu::wave_id()].val[buf][sl]);
}
// gfx950 row swaps: returns {new first operand, new second operand}
struct emu_uint2v { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
// v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second
static inline emu_uint2v __builtin_amdgcn_permlane16_swap(unsigned old, unsigned src, bool, bool) {
  int buf = emu::wave_exchange(((uint64_t)src << 32) | old, true);
  auto& ws = emu::ctx().waves[emu::wave_id()];
  const unsigned l = emu::lane_id(), row = l >> 4;
  auto D = [&](unsigned lane) { return (unsigned)(ws.val[buf][lane] & 0xffffffffu); };
  auto S = [&](unsigned lane) { return (unsigned)(ws.val[buf][lane] >> 32); };
  emu_uint2v r;
  r.v[0] = (row & 1) ? S(l - 16) : D(l);   // vdst: odd rows receive the second operand's preceding even row
  r.v[1] = (row & 1) ? S(l) : D(l + 16);   // src : even rows receive the first operand's following odd row
  return r;
}
// v_permlane32_swap: upper 32 lanes of the first operand <-> lower 32 lanes of the second
static inline emu_uint2v __builtin_amdgcn_permlane32_swap(unsigned old, unsigned src, bool, bool) {
  int buf = emu::wave_exchange(((uint64_t)src << 32) | old, true);
  auto& ws = emu::ctx().waves[emu::wave_id()];
  const unsigned l = emu::lane_id();
  auto D = [&](unsigned lane) { return (unsigned)(ws.val[buf][lane] & 0xffffffffu); };
  auto S = [&](unsigned lane) { return (unsigned)(ws.val[buf][lane] >> 32); };
  emu_uint2v r;
  r.v[0] = l >= 32 ? S(l - 32) : D(l);
  r.v[1] = l >= 32 ? S(l) : D(l + 32);
  return r;
}
// v_mfma_f32_16x16x4_f32: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j,
// D[4 R + t][j] in element t of lane 16 R + j (CDNA3/4 ISA guide, matrix layouts).  All 64 lanes must be active.
typedef float emu_v4f __attribute__((vector_size(16)));
static inline emu_v4f __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_v4f c, int, int, int) {
  int buf = emu::wave_exchange(((uint64_t)emu::from_bits<unsigned>(emu::to_bits(b)) << 32) | emu::from_bits<unsigned>(emu::to_bits(a)), true);
  auto& ws = emu::ctx().waves[emu::wave_id()];
  auto A = [&](unsigned lane) { return emu::from_bits<float>(ws.val[buf][lane] & 0xffffffffu); };
  auto B = [&](unsigned lane) { return emu::from_bits<float>(ws.val[buf][lane] >> 32); };
  const unsigned l = emu::lane_id(), R = l >> 4, j = l & 15;
  emu_v4f d = c;
  for (unsigned t = 0; t < 4; ++t)
    for (unsigned k = 0; k < 4; ++k) d[t] += A(16 * k + 4 * R + t) * B(16 * k + j);
  return d;
}
// hardware-id registers (read by the measurement build's probes only): a made-up placement of 7 "CUs" x 2 "XCCs"
static inline unsigned __builtin_amdgcn_s_getreg(int simm16) {
  const unsigned b = blockIdx.x;
  return (simm16 & 63) == 20 ? (b & 1u) : (((b * 5u + 3u) % 7u) << 8);
}
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = (T)v; }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned long long __ballot(int pred);
static inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(1); }   // the fibers of a wave meet here
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }

static inline unsigned __float_as_uint(float f) { return emu::from_bits<unsigned>(emu::to_bits(f)); }
static inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long b) { double r; std::memcpy(&r, &b, 8); return r; }
static inline int __float_as_int(float f) { return emu::from_bits<int>(emu::to_bits(f)); }
static inline float __uint_as_float(unsigned u) { return emu::from_bits<float>(emu::to_bits(u)); }
static inline float __int_as_float(int u) { return emu::from_bits<float>(emu::to_bits(u)); }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __log2f(x) log2f(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
static inline float __builtin_amdgcn_exp2f(float a) { return exp2f(a); }
static inline float __saturatef(float a) { return fminf(fmaxf(a, 0.f), 1.f); }
using std::max;
using std::min;

// atomics: one OS thread, so plain read-modify-write is exact
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
// device-side unsafe float atomic add (hardware FP atomic) — same semantics here
static inline float unsafeAtomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
