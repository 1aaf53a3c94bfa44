// aq_postproc.hip -- post-processing of the posterior inclusion probabilities on the device:
//   assign_bFDR        R/summarise_output.R:207-223   Bayesian FDR of every (SNP, trait) pair: sort all p q PPIs in
//                                                     decreasing order (ties in original order), running mean of 1 - PPI
//   hotspot sizes      R/summarise_output.R:98-105, 177-182   rowSums(gam_vb > thres) or rowSums(mat_fdr < thres)
// so that 4-32 GB of PPIs need not travel to the host just to be thresholded.  Sort and scan are hipCUB's
// (rocPRIM radix sort: stable, also in the descending variant; the library is part of ROCm, no hand-written kernel
// beats it for a plain key sort), the rest are three small kernels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <string>
#include "../../include/atlasqtl_hip.h"

int aq_fail_ext(int code, const std::string &msg);   // atlasqtl_hip.hip

#define AQP_HIP(call)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) { rc = aq_fail_ext(AQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); goto done; } \
  } while (0)

__global__ void aq_k_iota_u32(uint32_t *v, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ void aq_k_one_minus(const double *__restrict__ x, double *__restrict__ y, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 1.0 - x[i];
}
// fdr[ind[i]] = cumsum(1 - ppi_ord)[i] / (i + 1)      R/summarise_output.R:213-216
__global__ void aq_k_bfdr_scatter(const double *__restrict__ cs, const uint32_t *__restrict__ ind, double *__restrict__ fdr,
                                  uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fdr[ind[i]] = cs[i] / (double)(i + 1);
}

// d_ppi, d_fdr: device vectors of len doubles (as.vector of the p x q matrix); d_fdr may alias nothing.
int aq_bfdr_device(const double *d_ppi, double *d_fdr, int64_t len) {
  int rc = AQ_OK;
  if (len <= 0) return AQ_OK;
  if (len >= ((int64_t)1 << 31)) return aq_fail_ext(AQ_ERR_UNSUPPORTED, "assign_bFDR on the device handles fewer than 2^31 entries per call");
  const uint32_t n = (uint32_t)len;
  double *kout = nullptr, *tmpd = nullptr;
  uint32_t *vin = nullptr, *vout = nullptr;
  void *tmp = nullptr;
  size_t tb_sort = 0, tb_scan = 0, tb = 0;
  const unsigned grid = (n + 255) / 256;
  AQP_HIP(hipMalloc((void **)&kout, (size_t)n * sizeof(double)));
  AQP_HIP(hipMalloc((void **)&tmpd, (size_t)n * sizeof(double)));
  AQP_HIP(hipMalloc((void **)&vin, (size_t)n * sizeof(uint32_t)));
  AQP_HIP(hipMalloc((void **)&vout, (size_t)n * sizeof(uint32_t)));
  AQP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb_sort, d_ppi, kout, vin, vout, (int)n));
  AQP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, tmpd, kout, (int)n));
  tb = tb_sort > tb_scan ? tb_sort : tb_scan;
  AQP_HIP(hipMalloc(&tmp, tb));
  hipLaunchKernelGGL(aq_k_iota_u32, dim3(grid), dim3(256), 0, 0, vin, n);
  AQP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tb, d_ppi, kout, vin, vout, (int)n));   // ind <- order(vec_ppi, decreasing = TRUE)
  hipLaunchKernelGGL(aq_k_one_minus, dim3(grid), dim3(256), 0, 0, kout, tmpd, n);
  AQP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tb, tmpd, kout, (int)n));                          // cumsum(1 - vec_ppi_ord)
  hipLaunchKernelGGL(aq_k_bfdr_scatter, dim3(grid), dim3(256), 0, 0, kout, vout, d_fdr, n);
  AQP_HIP(hipGetLastError());
  AQP_HIP(hipDeviceSynchronize());
done:
  if (kout) hipFree(kout);
  if (tmpd) hipFree(tmpd);
  if (vin) hipFree(vin);
  if (vout) hipFree(vout);
  if (tmp) hipFree(tmp);
  return rc;
}

// rs[j] = #{k : m[j,k] > thres} (lt == 0) or #{k : m[j,k] < thres} (lt == 1); m is p x q column-major
__global__ void aq_k_row_count(const double *__restrict__ m, int64_t *__restrict__ rs, int p, int q, double thres, int lt) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p) return;
  int64_t c = 0;
  for (int k = 0; k < q; k++) {
    double v = m[(size_t)j + (size_t)p * k];
    c += lt ? (v < thres) : (v > thres);
  }
  rs[j] = c;
}
int aq_row_count_device(const double *d_m, int64_t *d_rs, int p, int q, double thres, int lt) {
  hipLaunchKernelGGL(aq_k_row_count, dim3((p + 255) / 256), dim3(256), 0, 0, d_m, d_rs, p, q, thres, lt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, std::string("aq_k_row_count: ") + hipGetErrorString(e));
  return AQ_OK;
}
