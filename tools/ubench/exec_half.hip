// Micro-benchmark: does a wave64 VALU instruction issue faster on gfx950 when part of EXEC is zero?
// The composite kernels run a 64-lane quadrant body for a Gaussian that covers ~29 of the 64 pixels (useful-lane
// fraction 0.45); if a half-empty EXEC halved the issue time, culling at 8x4 half-quadrant granularity would pay.
// Same harness as valu_rate.hip; the instruction stream runs under a fixed EXEC mask set by s_mov_b64 before the loop.
//   hipcc --offload-arch=gfx950 -O3 -o exec_half exec_half.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND> __global__ __launch_bounds__(256) void k(float* out, float seed, long long* cycles, unsigned long long mask) {
  float a[UNROLL];
  v2f p[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1.f}; }
  const float m = seed * 0.999f, c = seed * 1e-3f;
  const v2f m2 = {m, m}, c2 = {c, c};
  unsigned long long saved;
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
      if (KIND == 4) asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %1" :: "v"(a[i]), "v"(m) : "s10", "s11");
      if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int KIND> double run(int waves_per_simd, unsigned long long mask) {
  const int blocks = 256 * waves_per_simd;
  float* out; long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&cyc, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, cyc, mask);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, cyc, mask);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
  const double insts_per_simd = (double)ITERS * UNROLL * waves_per_simd;
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // wall_clock64 ticks at 100 MHz
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return ms * 1e6 * ghz / insts_per_simd;
}

struct Mask { const char* name; unsigned long long bits; };
static const Mask MASKS[] = {
  {"all 64 lanes", ~0ull}, {"lanes 0-31 (EXEC_HI = 0)", 0xffffffffull}, {"lanes 32-63 (EXEC_LO = 0)", 0xffffffff00000000ull},
  {"lanes 0-15", 0xffffull}, {"rows 0 and 2", 0x0000ffff0000ffffull}, {"rows 0 and 1 half each", 0x00ff00ff00ff00ffull},
  {"one lane", 1ull},
};

template <int KIND> void sweep(const char* name) {
  for (int w : {4, 6}) {
    printf("%-22s waves/SIMD %d:", name, w);
    for (const Mask& mk : MASKS) printf("  %5.2f", run<KIND>(w, mk.bits));
    printf("\n");
  }
}

int main() {
  printf("cycles per wave64 instruction per SIMD, by EXEC mask; columns:");
  for (const Mask& mk : MASKS) printf(" [%s]", mk.name);
  printf("\n");
  sweep<0>("v_fma_f32"); sweep<1>("v_pk_fma_f32"); sweep<2>("v_exp_f32"); sweep<7>("v_rcp_f32"); sweep<3>("v_cndmask_b32 sgpr");
  sweep<4>("v_cmp_gt_f32 sgpr"); sweep<5>("v_add_f32_dpp"); sweep<6>("v_med3_f32");
  return 0;
}
