// Micro-benchmark: cycles per wave64 VALU instruction on one SIMD of gfx950, by instruction class.
// Settles what "VALU-issue-bound" means for the composite kernels: is a plain fp32 op 2 or 4 cycles per wave64, what do
// packed-fp32, transcendental, DPP and permlane-swap ops cost.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND> __global__ __launch_bounds__(256) void k(float* out, float seed, long long* cycles) {
  float a[UNROLL];
  v2f p[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.f}; }
  const float m = seed * 0.999f, c = seed * 1e-3f;
  const v2f m2 = {m, m}, c2 = {c, c};
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 4) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % UNROLL]));
      if (KIND == 6) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % UNROLL]));
      if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
      if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if (KIND == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(m) : "vcc");
      if (KIND == 11) asm volatile("v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
      if (KIND == 14) asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %1" :: "v"(a[i]), "v"(m) : "s10", "s11");
      if (KIND == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      if (KIND == 17) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, 48, %0" : "+v"(p[i]) : "v"(m) : "s10", "s11");
      if (KIND == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (KIND == 19) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 20) asm volatile("v_readlane_b32 s10, %0, 3" :: "v"(a[i]) : "s10");
      if (KIND == 21) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 22) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 23) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int KIND> void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd x 4 SIMDs / 4 waves per block)
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&cyc, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  const double insts_per_simd = (double)ITERS * UNROLL * waves_per_simd;
  // wall_clock64 ticks at 100 MHz; the cycle counter is the shader clock: clock = cycles / (ticks / 100 MHz)
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  printf("%-24s waves/SIMD %d: %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at %.2f GHz (oldest wave alone: %.2f cycles/instr)\n",
         name, waves_per_simd, ms, ms * 1e6 * ghz / insts_per_simd, ghz, (double)h[0] / ((double)ITERS * UNROLL));
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  for (int w : {1, 4, 8}) {
    run<0>("v_fma_f32", w); run<12>("v_add_f32", w); run<7>("v_mul_f32", w); run<1>("v_pk_fma_f32", w); run<8>("v_pk_mul_f32", w);
    run<2>("v_exp_f32", w); run<3>("v_rcp_f32", w);
    run<4>("v_add_f32_dpp quad_perm", w); run<11>("v_mov_b32_dpp row_ror", w); run<5>("v_permlane32_swap", w); run<6>("v_permlane16_swap", w);
    run<9>("v_cndmask_b32 vcc", w); run<10>("v_cmp_gt_f32 vcc", w); run<13>("v_cndmask_b32_e64 sgpr", w); run<14>("v_cmp_gt_f32_e64 sgpr", w);
    run<15>("v_max_f32", w); run<16>("v_mov_b32", w); run<17>("v_mad_u64_u32", w); run<18>("cndmask+add pair", w); run<19>("v_fmac_f32", w);
    run<20>("v_readlane_b32", w); run<21>("v_log_f32", w); run<22>("v_cvt_f16_f32", w); run<23>("v_pk_add_f32", w);
    printf("\n");
  }
  return 0;
}
