// Microbenchmark: f64 MFMA rate as a function of how long the kernel runs (burst vs sustained), all 256 CUs and 32 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int iters, double a0, double b0) {
  d4 acc[4];
  for (int i = 0; i < 4; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out; hipMalloc(&out, 512 * 512 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {256, 32}) {
    for (int iters : {500, 1000, 2000, 4000, 8000, 16000, 32000, 64000, 128000}) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {   // 8 waves per WG = 2 per SIMD
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, iters, 1.0, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double flop = (double)grid * 8 * iters * 4 * 2048.0;
      printf("%3d WGs x 8 waves, %6d x 4 MFMAs per wave: %8.3f ms  %.1f TFLOP/s (%.1f per-256-CU equivalent)\n", grid, iters, ms,
             flop / ms / 1e9, flop / ms / 1e9 * 256 / grid);
    }
  }
  return 0;
}
