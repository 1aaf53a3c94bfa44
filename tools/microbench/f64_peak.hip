// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950.
// (MI355X_MICROARCH.md has no FP64 row; SURVEY section 8d asks for a measured peak.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-9 + i;
  double a = a0, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *out; hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  int iters = 20000;
  for (int wpc : {1, 2, 4}) {   // workgroups (of 4 waves) per CU
    int grid = 256 * wpc;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    double flop = (double)grid * 4 * iters * 4 * 2048.0;
    printf("mfma_f64_16x16x4 NACC=4  %d WG/CU: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpc, ms, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 4.0 * wpc));
  }
  {
    int grid = 256;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
    printf("mfma_f64_16x16x4 NACC=1 (dependent chain) 1 WG/CU: %.3f ms  %.1f cycles/MFMA @2.4GHz\n", ms, ms * 1e-3 * 2.4e9 / iters);
  }
  for (int wpc : {1, 2, 4}) {
    int grid = 256 * wpc;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    double flop = (double)grid * 256 * iters * 8 * 2.0;
    printf("v_fma_f64 NACC=8  %d WG/CU: %.3f ms  %.2f TFLOP/s\n", wpc, ms, flop / ms / 1e9);
  }
  {
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fma<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    printf("v_fma_f64 dependent chain: %.1f cycles per FMA @2.4GHz\n", ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
