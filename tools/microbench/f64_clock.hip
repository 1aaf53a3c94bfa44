// Microbenchmark: what clock does the chip sustain under f64 MFMA load, and how many shader cycles does one
// v_mfma_f64_16x16x4_f64 take?  s_memtime (clock64) counts shader-clock cycles, s_memrealtime (wall_clock64) a
// constant 100 MHz: their ratio is the clock.  Run for different numbers of active workgroups (one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, long long *clk, int iters, double a0, double b0, int waves) {
  if ((int)(threadIdx.x >> 6) >= waves) return;
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  double *out; hipMalloc(&out, 1024 * 256 * sizeof(double));
  long long *clk; hipMalloc(&clk, 2 * 1024 * sizeof(long long));
  int iters = 20000;
  std::vector<long long> h(2 * 1024);
  printf("grid waves/WG : shader cycles per MFMA per wave | clock GHz (s_memtime / s_memrealtime@100MHz) | TFLOP/s\n");
  for (int waves : {4, 8 /* two WGs per CU */}) {
    for (int grid : {32, 64, 128, 256}) {
      int g = waves == 8 ? grid * 2 : grid;
      int w = 4;
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_mfma<4>, dim3(g), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-3, w);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), clk, 2 * g * sizeof(long long), hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < g; i++) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
      cyc /= g; wall /= g;
      double ghz = cyc / (wall * 10.0);   // wall ticks are 10 ns
      double per = cyc / (iters * 4.0);
      double tf = (double)g * w * iters * 4 * 2048.0 / (wall * 10e-9) / 1e12;
      printf("%4d WGs (%d per CU, 4 waves each): %.1f cycles/MFMA/wave, clock %.3f GHz, %.2f TFLOP/s\n", g, waves / 4, per, ghz, tf);
    }
  }
  // one wave per SIMD with a dependent chain: MFMA latency in cycles
  hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256), 0, 0, out, clk, iters, 1.0, 1e-3, 4);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), clk, 2 * 256 * sizeof(long long), hipMemcpyDeviceToHost);
  printf("dependent chain: %.1f cycles per MFMA, clock %.3f GHz\n", (double)h[0] / iters, (double)h[0] / (h[1] * 10.0));
  return 0;
}
