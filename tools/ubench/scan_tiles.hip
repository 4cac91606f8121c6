// Micro-benchmark: where the single-workgroup tile scan (binning.hip k_scan_tiles) spends its time at 1080p (8160 tiles).
// Includes the product kernel source unchanged with GS_SCAN_PROBE: thread 0 stamps the 100 MHz wall clock at phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -I instantsplat_amd/csrc -o tools/ubench/scan_tiles tools/ubench/scan_tiles.hip
#define GS_SCAN_PROBE 1
#include "binning.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_touch(int T, uint32_t* count, const uint32_t* src) {   // the state the count kernel leaves: counts written by atomics
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < T) { count[i] = 0; atomicAdd(&count[i], src[i]); }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8160;
  std::vector<uint32_t> h(T);
  srand(7);
  for (int i = 0; i < T; ++i) { const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    const double g = std::sqrt(-2 * std::log(u)) * std::cos(6.283185307 * v); h[i] = (i % 17 == 0) ? 0u : (uint32_t)(600.0 * std::exp(0.8 * g)); }
  uint32_t *src, *count, *start, *order, *meta, *seg, *part; int32_t* nr;
  CHECK(hipMalloc(&src, T * 4)); CHECK(hipMalloc(&count, (T + 1) * 4)); CHECK(hipMalloc(&start, (T + 1) * 4)); CHECK(hipMalloc(&order, T * 4));
  CHECK(hipMalloc(&meta, 64)); CHECK(hipMalloc(&seg, (T + 1) * 4)); CHECK(hipMalloc(&part, (T + 1) * 4)); CHECK(hipMalloc(&nr, 4));
  CHECK(hipMemcpy(src, h.data(), T * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int staged = 0; staged < 2; ++staged) {
    const size_t stage_bytes = (size_t)((T + SCAN_THREADS - 1) / SCAN_THREADS) * SCAN_THREADS * 4;
    if (staged && !(T > SCAN_THREADS && stage_bytes <= SCAN_STAGE_MAX_BYTES)) continue;
    double tot = 0; long long ph[16] = {0}; const int reps = 50;
    for (int r = 0; r < reps + 5; ++r) {
      hipLaunchKernelGGL(k_touch, dim3((T + 255) / 256), dim3(256), 0, 0, T, count, src);
      CHECK(hipEventRecord(e0, 0));
      if (staged) hipLaunchKernelGGL(k_scan_tiles<true>, dim3(1), dim3(SCAN_THREADS), stage_bytes, 0, T, count, start, nr, order, meta, seg, part, 1024u);
      else hipLaunchKernelGGL(k_scan_tiles<false>, dim3(1), dim3(SCAN_THREADS), 0, 0, T, count, start, nr, order, meta, seg, part, 1024u);
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      long long p[16]; CHECK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_scan_probe), sizeof(p)));
      if (r >= 5) { tot += ms; for (int k = 1; k <= 8; ++k) ph[k] += p[k] - p[k - 1]; }
    }
    printf("T %d staged %d: event %.2f us per launch; phases (us): stage %.2f sum+scan %.2f max %.2f hist %.2f histscan %.2f order %.2f units %.2f start %.2f | kernel body %.2f\n",
           T, staged, tot / reps * 1e3, ph[1] / reps * 0.01, ph[2] / reps * 0.01, ph[3] / reps * 0.01, ph[4] / reps * 0.01, ph[5] / reps * 0.01, ph[6] / reps * 0.01,
           ph[7] / reps * 0.01, ph[8] / reps * 0.01, (ph[1]+ph[2]+ph[3]+ph[4]+ph[5]+ph[6]+ph[7]+ph[8]) / reps * 0.01);
  }
  return 0;
}
