// Micro-benchmark (gfx950): do fp64 VALU operations and v_mfma_f64_16x16x4_f64 share an execution resource?
//   hipcc --offload-arch=gfx950 -O3 -o ubench_f64_pipe tools/ubench_f64_pipe.hip && ./ubench_f64_pipe
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD, as in aq_core_sweep_la.h).  Waves 0-3 stream MFMAs (two
// independent accumulators), waves 4-7 stream a second kind of work (nothing / fp64 FMA / fp32 FMA / int32 multiply-add / MFMA),
// each for a fixed instruction count; the kernel's duration tells whether the two streams overlap or add up.
// The look-ahead kernel's roofline argument in DESIGN.md section 6 rests on the numbers this prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND_A, int KIND_B>   // 0 nothing, 1 MFMA f64, 2 FMA f64, 3 FMA f32, 4 mad i32
__global__ __launch_bounds__(512) void k_pipe(double *out, int iters, double seed) {
  const int w = threadIdx.x >> 6;
  const int kind = (w < 4) ? KIND_A : KIND_B;
  double r = 0.0;
  if (kind == 1) {
    d4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    const double x = seed, y = seed * 0.5;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
      }
    }
    r = a0[0] + a1[1];
  } else if (kind == 2) {
    double c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = seed + u;
    const double m = 1.0 - 1e-9 * seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int rep = 0; rep < 16; rep++)     // 16 x 8 = 128 FMAs per iteration (the MFMA stream: 16 per iteration)
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = __builtin_fma(c[u], m, seed);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) r += c[u];
  } else if (kind == 3) {
    float c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = (float)seed + u;
    const float m = 1.0f - 1e-6f * (float)seed, s = (float)seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int rep = 0; rep < 16; rep++)
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = __builtin_fmaf(c[u], m, s);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) r += c[u];
  } else if (kind == 4) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) c[u] = (int)seed + u + threadIdx.x;
    const int m = 3 + (int)seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int rep = 0; rep < 16; rep++)
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = c[u] * m + u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) r += c[u];
  }
  if (r == 12345.678) out[threadIdx.x] = r;   // keeps the streams alive
}

template <int A, int B>
static double run(const char *name, double *out, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_pipe<A, B><<<256, 512>>>(out, iters / 8, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_pipe<A, B><<<256, 512>>>(out, iters, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-46s %8.3f ms\n", name, ms);
  return ms;
}

int main() {
  double *out;
  CHECK(hipMalloc(&out, 4096));
  const int iters = 20000;   // per wave: 320 000 MFMAs (16 / iteration) or 2 560 000 VALU operations (128 / iteration)
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, 0));
  printf("# %s, %d CUs, clock %d MHz; 256 workgroups x 8 waves; waves 0-3 = stream A, waves 4-7 (same SIMDs) = stream B\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
  printf("# per wave: A = %d MFMA f64 16x16x4, B = %d VALU operations (or the same MFMA count)\n", iters * 16, iters * 128);
  const double a = run<1, 0>("A: MFMA f64 alone", out, iters);
  const double b2 = run<0, 2>("B: FMA f64 alone", out, iters);
  const double b3 = run<0, 3>("B: FMA f32 alone (v_pk_fma_f32, 2 per op)", out, iters);
  const double b4 = run<0, 4>("B: mad i32 alone (v_mad_u64_u32 / mul_lo)", out, iters);
  const double ab1 = run<1, 1>("A + B: MFMA f64 + MFMA f64", out, iters);
  const double ab2 = run<1, 2>("A + B: MFMA f64 + FMA f64", out, iters);
  const double ab3 = run<1, 3>("A + B: MFMA f64 + FMA f32 (packed)", out, iters);
  const double ab4 = run<1, 4>("A + B: MFMA f64 + mad i32", out, iters);
  printf("# cycles per MFMA (alone): %.1f   per FMA f64: %.2f   per FMA f32: %.2f   per mad i32: %.2f   (at the reported clock)\n",
         a * 1e-3 * pr.clockRate * 1e3 / (iters * 16.0), b2 * 1e-3 * pr.clockRate * 1e3 / (iters * 128.0),
         b3 * 1e-3 * pr.clockRate * 1e3 / (iters * 128.0), b4 * 1e-3 * pr.clockRate * 1e3 / (iters * 128.0));
  printf("# overlap = (A + B alone - together) / min(A, B):  MFMA+MFMA %.2f   MFMA+FMA f64 %.2f   MFMA+FMA f32 %.2f   MFMA+mad i32 %.2f\n",
         (2 * a - ab1) / a, (a + b2 - ab2) / (a < b2 ? a : b2), (a + b3 - ab3) / (a < b3 ? a : b3), (a + b4 - ab4) / (a < b4 ? a : b4));
  return 0;
}
