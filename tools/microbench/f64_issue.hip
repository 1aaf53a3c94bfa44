// Microbenchmark: f64 MFMA issue rate per SIMD as a function of waves per SIMD, dependent vs independent accumulation,
// and 3 vs 4 active SIMDs per CU (the sweep kernel keeps SIMD 3 free of MFMAs for the recurrence wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double *out, long long *clk, int iters, double a0, double b0, int simd3_off) {
  const int w = threadIdx.x >> 6;
  if (simd3_off && (w & 3) == 3) return;
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  __syncthreads();
  long long c0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + w] = c1 - c0;
}

template <int NACC>
static void run(double *out, long long *clk, int wps, int simd3_off) {
  const int iters = 8000 / NACC * 4;
  const int waves = 4 * wps;
  std::vector<long long> h(256 * 16);
  hipMemset(clk, 0, 256 * 16 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(64 * waves), 0, 0, out, clk, iters, 1.0, 1e-3, simd3_off);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(h.data(), clk, 256 * 16 * sizeof(long long), hipMemcpyDeviceToHost);
  double cyc = 0; int cnt = 0;
  for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) if (!(simd3_off && (w & 3) == 3)) { cyc += h[b * 16 + w]; cnt++; }
  cyc /= cnt;
  double per_wave = cyc / ((double)iters * NACC);
  const int nsimd = simd3_off ? 3 : 4;
  const double wall_ns_per_mfma_simd = ms * 1e6 / ((double)iters * NACC * wps);
  printf("%d waves/SIMD, %s accumulators, %d SIMDs: s_memtime %.1f ticks/MFMA/wave; wall clock %.1f ns per MFMA and SIMD (%.1f cycles @2.39 GHz), %.1f TFLOP/s\n",
         wps, NACC == 1 ? "1 (dependent)" : NACC == 2 ? "2" : "4 (independent)", nsimd, per_wave, wall_ns_per_mfma_simd,
         wall_ns_per_mfma_simd * 2.39, 256.0 * nsimd * 2048.0 / wall_ns_per_mfma_simd / 1e3);
}

int main() {
  double *out; hipMalloc(&out, 256 * 1024 * sizeof(double));
  long long *clk; hipMalloc(&clk, 256 * 16 * sizeof(long long));
  for (int off : {0, 1})
    for (int wps : {1, 2, 3, 4}) {
      run<1>(out, clk, wps, off);
      run<2>(out, clk, wps, off);
      run<4>(out, clk, wps, off);
    }
  return 0;
}
