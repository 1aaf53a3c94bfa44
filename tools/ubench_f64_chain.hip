// Micro-benchmark (gfx950): how fast can ONE wave stream v_mfma_f64_16x16x4_f64 when the MFMAs form one, two or four chains through
// the accumulator (SrcC = the previous result), and what do two such waves on one SIMD reach together?
//   hipcc --offload-arch=gfx950 -O3 -o ub tools/ubench_f64_chain.hip && ./ub
// One-tile workgroups of the sweep kernel issue 4 dependent MFMAs per residual tile and trait tile (update), then 4 more (S').
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NCH, int WAVES>   // NCH accumulator chains per wave, WAVES waves per SIMD (workgroup = 4 * WAVES waves)
__global__ __launch_bounds__(256 * WAVES) void k_chain(double *out, int iters, double seed) {
  d4 a[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) a[c] = (d4){0, 0, 0, 0};
  const double x = seed, y = seed * 0.5;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16 / NCH; u++)
#pragma unroll
      for (int c = 0; c < NCH; c++) a[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[c], 0, 0, 0);
  }
  double r = 0.0;
#pragma unroll
  for (int c = 0; c < NCH; c++) r += a[c][0];
  if (r == 12345.678) out[threadIdx.x] = r;
}

template <int NCH, int WAVES>
static void run(double *out, int iters, int clock_khz) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_chain<NCH, WAVES><<<256, 256 * WAVES>>>(out, iters / 8, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_chain<NCH, WAVES><<<256, 256 * WAVES>>>(out, iters, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms * 1e-3 * clock_khz * 1e3 / (iters * 16.0 * WAVES);   // SIMD cycles per MFMA
  printf("%d chain(s) per wave, %d wave(s) per SIMD: %8.3f ms  %6.1f SIMD-cycles per MFMA  (%.0f %% of the 64-cycle rate)\n", NCH, WAVES, ms, cyc, 6400.0 / cyc);
}

int main() {
  double *out;
  CHECK(hipMalloc(&out, 8192));
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  const int iters = 20000;
  printf("# 256 workgroups (one per CU), every wave issues %d MFMAs\n", iters * 16);
  run<1, 1>(out, iters, pr.clockRate); run<2, 1>(out, iters, pr.clockRate); run<4, 1>(out, iters, pr.clockRate);
  run<1, 2>(out, iters, pr.clockRate); run<2, 2>(out, iters, pr.clockRate); run<4, 2>(out, iters, pr.clockRate);
  return 0;
}
