// Micro-benchmark: do the matrix pipe and the VALU of one gfx950 SIMD overlap for the instruction mix of the backward
// composite kernel?  A wave runs ITERS rounds of [NV plain fp32 VALU instructions, then NM dependent v_mfma_f32_16x16x4_f32
// on one accumulator, then the accumulator read-back] — the shape of one (Gaussian, tile) step whose cross-row reduction runs
// on the matrix pipe (composite.hip, GS_BW_MFMA build).  Reported: cycles per round per SIMD with W waves resident per SIMD,
// for the VALU part alone, the MFMA part alone and both.  If the pipes overlapped perfectly "both" would equal the larger of
// the two; if they serialised, their sum.     hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 1024;
typedef float v4f __attribute__((ext_vector_type(4)));

// SHAPE 0: VALU block, then NM MFMAs chained on ONE accumulator (the GS_BW_MFMA build of composite.hip)
// SHAPE 1: VALU block, then NM MFMAs on NM / 3 chains of three (independent accumulators)
// SHAPE 2: the NM MFMAs (independent accumulators) spread evenly through the VALU block
template <int NV, int NM, int KIND, int SHAPE = 0> __global__ __launch_bounds__(64) void k(float* out, float seed, long long* cycles) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
  const float m = seed * 0.999f, c = seed * 1e-3f;
  const float sel = (threadIdx.x & 15) == 3 ? 1.f : 0.f;
  float total = 0.f;
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
  auto mfma = [&](v4f acc, float x) {
    asm volatile("" : "+v"(x));   // loop-variant as far as the compiler knows: nothing is hoisted out of the round
    if (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(x, sel, acc, 0, 0, 0);
    v8bf p, q;
    asm volatile("" : "=v"(p), "=v"(q));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, acc, 0, 0, 0);
  };
  for (int it = 0; it < ITERS; ++it) {
    constexpr int NACC = SHAPE == 0 ? 1 : (NM >= 3 ? NM / 3 : 1);
    v4f acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};
    int issued = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(m), "v"(c));
      if (SHAPE == 2 && NM > 0 && (i + 1) % (NV / NM) == 0 && issued < NM) { acc[issued % NACC] = mfma(acc[issued % NACC], a[i & 7]); ++issued; }
    }
    if (SHAPE != 2 || NV == 0) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i % NACC] = mfma(acc[i % NACC], a[i & 7]);
    }
    if (NM > 0) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) total += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  float s = total;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int NV, int NM, int KIND, int SHAPE = 0> double run(const char* name, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd;  // single-wave workgroups, like k_composite_bwd
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 64));
  CHECK(hipMalloc(&cyc, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NV, NM, KIND, SHAPE>), dim3(blocks), dim3(64), 0, 0, out, 1.0f, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NV, NM, KIND, SHAPE>), dim3(blocks), dim3(64), 0, 0, out, 1.0f, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  const double per_round = ms * 1e6 * ghz / ((double)ITERS * waves_per_simd);
  printf("%-44s waves/SIMD %d: %8.3f ms -> %7.1f cycles per round per SIMD at %.2f GHz\n", name, waves_per_simd, ms, per_round, ghz);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
  return per_round;
}

int main() {
  for (int w : {1, 4, 6}) {
    // one backward step of the shipped kernel is ~343 issue cycles, 128 of them the reduction; with the cross-row half on the
    // matrix pipe ~140 plain-VALU-equivalents (280 cycles) remain next to 9 MFMAs
    const double v = run<140, 0, 0>("140 v_fma_f32", w);
    const double m = run<0, 9, 0>("9 v_mfma_f32_16x16x4_f32, one chain", w);
    run<0, 9, 0, 1>("9 v_mfma_f32_16x16x4_f32, three chains", w);
    const double b = run<140, 9, 0>("140 v_fma + 9 mfma f32 (one chain, at the end)", w);
    const double b1 = run<140, 9, 0, 1>("140 v_fma + 9 mfma f32 (three chains, at the end)", w);
    const double b2 = run<140, 9, 0, 2>("140 v_fma + 9 mfma f32 (three chains, interleaved)", w);
    printf("   -> f32: extra cycles per MFMA next to the VALU block: one chain %.1f, three chains %.1f, interleaved %.1f (pipe time 32); alone %.1f\n",
           (b - v) / 9, (b1 - v) / 9, (b2 - v) / 9, m / 9);
    const double m16 = run<0, 12, 1, 1>("12 v_mfma_f32_16x16x32_bf16, four chains", w);
    const double c0 = run<140, 12, 1>("140 v_fma + 12 mfma bf16 (one chain, at the end)", w);
    const double c1 = run<140, 12, 1, 1>("140 v_fma + 12 mfma bf16 (four chains, at the end)", w);
    const double c2 = run<140, 12, 1, 2>("140 v_fma + 12 mfma bf16 (four chains, interleaved)", w);
    printf("   -> bf16 16x16x32: extra cycles per MFMA: one chain %.1f, four chains %.1f, interleaved %.1f (pipe time 16); alone %.1f\n\n",
           (c0 - v) / 12, (c1 - v) / 12, (c2 - v) / 12, m16 / 12);
  }
  return 0;
}
