// Microbenchmark: latency (dependent chain) and throughput (8 independent chains) of the fp64 VALU operations that
// make up the recurrence's dependency chain; one wave per SIMD, s_memtime ticks (= shader clock here) per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int NACC>
__global__ void k(double *out, long long *clk, int iters, double a, double b) {
  double x[NACC];
  for (int i = 0; i < NACC; i++) x[i] = 1.0 + threadIdx.x * 1e-9 + i;
  long long c0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 16; rep++)      // 16 steps per loop iteration: the loop overhead is amortised
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
      if (OP == 1) x[i] = x[i] * a;
      if (OP == 2) x[i] = x[i] + b;
      if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]) + b;     // rcp + add (keeps the value in range)
      if (OP == 4) x[i] = __builtin_rint(x[i] * a);
      if (OP == 5) x[i] = __builtin_ldexp(x[i], 1) * a;
      if (OP == 6) x[i] = __builtin_fmin(x[i] * a, 800.0);
    }
  }
  long long c1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
template <int OP, int NACC>
static void run(const char *name, int nops, double *out, long long *clk, double a, double b) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<OP, NACC>), dim3(64), dim3(256), 0, 0, out, clk, iters, a, b); hipDeviceSynchronize(); }
  long long h; hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s %d chain(s): %.1f ticks per %d-op step per chain-set -> %.1f ticks per op (issue+latency)\n", name, NACC,
         (double)h / iters / 16, nops, (double)h / iters / 16 / (nops * (NACC == 1 ? 1 : NACC)));
}
int main() {
  double *out; hipMalloc(&out, 64 * 256 * sizeof(double));
  long long *clk; hipMalloc(&clk, 64 * sizeof(long long));
  run<0, 1>("v_fma_f64", 1, out, clk, 1.0000001, 1e-9);   run<0, 8>("v_fma_f64", 1, out, clk, 1.0000001, 1e-9);
  run<1, 1>("v_mul_f64", 1, out, clk, 1.0000001, 0);      run<1, 8>("v_mul_f64", 1, out, clk, 1.0000001, 0);
  run<2, 1>("v_add_f64", 1, out, clk, 0, 1e-9);           run<2, 8>("v_add_f64", 1, out, clk, 0, 1e-9);
  run<3, 1>("v_rcp_f64 + add", 2, out, clk, 0, 1.0);      run<3, 8>("v_rcp_f64 + add", 2, out, clk, 0, 1.0);
  run<4, 1>("v_mul + v_rndne", 2, out, clk, 1.0000001, 0); run<4, 8>("v_mul + v_rndne", 2, out, clk, 1.0000001, 0);
  run<5, 1>("v_ldexp + v_mul", 2, out, clk, 0.5, 0);      run<5, 8>("v_ldexp + v_mul", 2, out, clk, 0.5, 0);
  run<6, 1>("v_mul + v_min", 2, out, clk, 1.0000001, 0);  run<6, 8>("v_mul + v_min", 2, out, clk, 1.0000001, 0);
  return 0;
}
