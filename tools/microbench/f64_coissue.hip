// Microbenchmark: does VALU work issued between f64 MFMAs of the same wave slow the MFMA stream?  Three variants per
// MFMA: nothing, 12 integer selects (v_cndmask_b32 / v_bfe_u32, as a 2-bit genotype decode would need), 6 fp64 FMAs.
// Two waves per SIMD, 256 workgroups; wall-clock ns per MFMA and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double *out, int iters, double a0, double b0, unsigned code) {
  d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned c = code + threadIdx.x;
  double t0 = 0.5, t1 = 1.5, t2 = -0.5, f = 1.0;
  unsigned long long sel = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 3; r++) {   // 3 decoded values: bfe + 2 compares + 4 cndmask each
          unsigned g = (c >> (2 * r)) & 3u;
          double v = g == 0 ? t0 : (g == 1 ? t1 : t2);
          sel += (unsigned long long)__double_as_longlong(v) >> 60;
        }
        c = c * 1664525u + 1013904223u;
      }
      if (MODE == 2) {
#pragma unroll
        for (int r = 0; r < 6; r++) f = __builtin_fma(f, 1.0000001, 1e-9);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + f + (double)sel;
}
template <int MODE>
static void run(const char *name, double *out) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0, 1e-3, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-28s %.1f ns per MFMA and SIMD (%.1f TFLOP/s)\n", name, ms * 1e6 / (iters * 2.0 * 2), 256.0 * 8 * iters * 2 * 2048.0 / ms / 1e9);
}
int main() {
  double *out; hipMalloc(&out, 256 * 512 * sizeof(double));
  run<0>("MFMA only", out);
  run<1>("MFMA + 3 integer decodes", out);
  run<2>("MFMA + 6 fp64 FMAs", out);
  return 0;
}
