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
#include <stdlib.h>
#include <string>
#include "../../include/atlasqtl_hip.h"

int aq_fail_ext(int code, const std::string &msg);   // atlasqtl_hip.hip

#define AQP_HIP(call)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) { rc = aq_fail_ext(AQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); goto done; } \
  } while (0)

template <typename I>
__global__ void aq_k_iota(I *v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (I)i;
}
__global__ void aq_k_one_minus(const double *__restrict__ x, double *__restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 1.0 - x[i];
}
// fdr[ind[i]] = cumsum(1 - ppi_ord)[i] / (i + 1)      R/summarise_output.R:213-216
template <typename I>
__global__ void aq_k_bfdr_scatter(const double *__restrict__ cs, const I *__restrict__ ind, double *__restrict__ fdr, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fdr[ind[i]] = cs[i] / (double)(i + 1);
}

// Descending stable sort of the PPIs with their positions and the running sum of 1 - PPI along that order: the common part
// of assign_bFDR and of the sharded cutoff search below.  Index type: 32 bits below 2^32 entries, 64 bits beyond (the p q
// of C5 on one GPU is 4e9).  Allocates *keys (sorted PPIs), *csum (inclusive running sum of 1 - PPI), *idx (original
// position of each sorted entry); the caller frees them.
template <typename I>
static int aq_sort_ppi(const double *d_ppi, size_t n, double **keys, double **csum, I **idx) {
  int rc = AQ_OK;
  double *tmpd = nullptr;
  I *vin = nullptr;
  void *tmp = nullptr;
  size_t tb_sort = 0, tb_scan = 0, tb = 0;
  const unsigned grid = (unsigned)((n + 255) / 256);
  *keys = nullptr; *csum = nullptr; *idx = nullptr;
  AQP_HIP(hipMalloc((void **)keys, n * sizeof(double)));
  AQP_HIP(hipMalloc((void **)csum, n * sizeof(double)));
  AQP_HIP(hipMalloc((void **)&tmpd, n * sizeof(double)));
  AQP_HIP(hipMalloc((void **)&vin, n * sizeof(I)));
  AQP_HIP(hipMalloc((void **)idx, n * sizeof(I)));
  AQP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb_sort, d_ppi, *keys, vin, *idx, (int64_t)n));
  AQP_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, tmpd, *csum, (int64_t)n));
  tb = tb_sort > tb_scan ? tb_sort : tb_scan;
  AQP_HIP(hipMalloc(&tmp, tb));
  hipLaunchKernelGGL((aq_k_iota<I>), dim3(grid), dim3(256), 0, 0, vin, n);
  AQP_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tb, d_ppi, *keys, vin, *idx, (int64_t)n));   // ind <- order(vec_ppi, decreasing = TRUE)
  hipLaunchKernelGGL(aq_k_one_minus, dim3(grid), dim3(256), 0, 0, *keys, tmpd, n);
  AQP_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tb, tmpd, *csum, (int64_t)n));                           // cumsum(1 - vec_ppi_ord)
  AQP_HIP(hipGetLastError());
  AQP_HIP(hipDeviceSynchronize());
done:
  if (tmpd) hipFree(tmpd);
  if (vin) hipFree(vin);
  if (tmp) hipFree(tmp);
  if (rc != AQ_OK) {
    if (*keys) hipFree(*keys);
    if (*csum) hipFree(*csum);
    if (*idx) hipFree(*idx);
    *keys = *csum = nullptr; *idx = nullptr;
  }
  return rc;
}

template <typename I>
static int aq_bfdr_typed(const double *d_ppi, double *d_fdr, size_t n) {
  double *keys = nullptr, *csum = nullptr;
  I *idx = nullptr;
  int rc = aq_sort_ppi<I>(d_ppi, n, &keys, &csum, &idx);
  if (rc != AQ_OK) return rc;
  hipLaunchKernelGGL((aq_k_bfdr_scatter<I>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, csum, idx, d_fdr, n);
  if (hipDeviceSynchronize() != hipSuccess) rc = aq_fail_ext(AQ_ERR_DEVICE, "assign_bFDR: scatter failed");
  hipFree(keys); hipFree(csum); hipFree(idx);
  return rc;
}

// d_ppi, d_fdr: device vectors of len doubles (as.vector of the p x q matrix); d_fdr may alias nothing.
static bool aq_force_idx64() { const char *e = getenv("AQ_BFDR_IDX64"); return e && e[0] == '1'; }   // test hook: 64-bit positions at any size
int aq_bfdr_device(const double *d_ppi, double *d_fdr, int64_t len) {
  if (len <= 0) return AQ_OK;
  if ((uint64_t)len < (1ull << 32) && !aq_force_idx64()) return aq_bfdr_typed<uint32_t>(d_ppi, d_fdr, (size_t)len);
  return aq_bfdr_typed<uint64_t>(d_ppi, d_fdr, (size_t)len);
}

// ---- Bayesian FDR under trait sharding --------------------------------------------------------------------------------
// assign_bFDR ranks ALL p q PPIs (R/summarise_output.R:207-223).  With the traits spread over ranks, a rank sorts its own
// shard once (aq_shard_sort) and then answers, for a candidate PPI value c, "how many of my entries are >= c / > c and what
// is the sum of 1 - PPI over them" (aq_shard_query: two binary searches on the sorted shard).  Summed over the ranks that
// gives the running mean at the end of c's tie block; the running mean is non-decreasing along the order, so
// {FDR < thres} is a prefix of it and a bisection over c (one 4-double all-reduce per step, driven by the caller) finds it.
struct aq_shard_sorted {
  double *keys = nullptr, *csum = nullptr;
  uint32_t *idx32 = nullptr;
  uint64_t *idx64 = nullptr;
  size_t n = 0;
};
void aq_shard_free(aq_shard_sorted *s) {
  if (!s) return;
  if (s->keys) hipFree(s->keys);
  if (s->csum) hipFree(s->csum);
  if (s->idx32) hipFree(s->idx32);
  if (s->idx64) hipFree(s->idx64);
  delete s;
}
int aq_shard_sort(const double *d_ppi, int64_t len, aq_shard_sorted **out) {
  aq_shard_sorted *s = new aq_shard_sorted();
  s->n = (size_t)len;
  int rc = ((uint64_t)len < (1ull << 32) && !aq_force_idx64()) ? aq_sort_ppi<uint32_t>(d_ppi, s->n, &s->keys, &s->csum, &s->idx32)
                                        : aq_sort_ppi<uint64_t>(d_ppi, s->n, &s->keys, &s->csum, &s->idx64);
  if (rc != AQ_OK) { delete s; return rc; }
  *out = s;
  return AQ_OK;
}
// out[0] = #{ppi >= c}, out[1] = sum(1 - ppi : ppi >= c), out[2] = #{ppi > c}, out[3] = sum(1 - ppi : ppi > c),
// out[4] = the largest ppi < c (or -1): the value of the next tie block down
__global__ void aq_k_shard_query(const double *__restrict__ keys, const double *__restrict__ csum, size_t n, double c, double *out) {
  // keys is descending: first position with key < c, first position with key <= c
  size_t lo = 0, hi = n;
  while (lo < hi) { size_t m = lo + (hi - lo) / 2; if (keys[m] >= c) lo = m + 1; else hi = m; }
  const size_t nge = lo;
  lo = 0; hi = nge;
  while (lo < hi) { size_t m = lo + (hi - lo) / 2; if (keys[m] > c) lo = m + 1; else hi = m; }
  const size_t ngt = lo;
  out[0] = (double)nge; out[1] = nge ? csum[nge - 1] : 0.0;
  out[2] = (double)ngt; out[3] = ngt ? csum[ngt - 1] : 0.0;
  out[4] = nge < n ? keys[nge] : -1.0;
}
int aq_shard_query(const aq_shard_sorted *s, double c, double out[5]) {
  double *d = nullptr;
  if (hipMalloc((void **)&d, 5 * sizeof(double)) != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, "aq_shard_query: hipMalloc failed");
  hipLaunchKernelGGL(aq_k_shard_query, dim3(1), dim3(1), 0, 0, s->keys, s->csum, s->n, c, d);
  hipError_t e = hipMemcpy(out, d, 5 * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, std::string("aq_shard_query: ") + hipGetErrorString(e));
  return AQ_OK;
}
// rs[j] += 1 for every entry of the first `upto` sorted positions (column-major position -> row = position % p), and for the
// `take` entries of the tie block [t0, t1) that come first in the original order (the sort is stable: block order = index order)
template <typename I>
__global__ void aq_k_shard_rows(const I *__restrict__ idx, size_t upto, size_t t0, size_t take, int p, unsigned long long *rs) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < upto) atomicAdd(&rs[(size_t)idx[i] % (size_t)p], 1ull);
  else if (i - upto < take) atomicAdd(&rs[(size_t)idx[t0 + (i - upto)] % (size_t)p], 1ull);
}
int aq_shard_rows(const aq_shard_sorted *s, int64_t upto, int64_t t0, int64_t take, int p, int64_t *rs_host) {
  unsigned long long *d = nullptr;
  if (hipMalloc((void **)&d, (size_t)p * sizeof(unsigned long long)) != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, "aq_shard_rows: hipMalloc failed");
  hipMemset(d, 0, (size_t)p * sizeof(unsigned long long));
  const size_t tot = (size_t)upto + (size_t)take;
  if (tot > 0) {
    const unsigned grid = (unsigned)((tot + 255) / 256);
    if (s->idx32) hipLaunchKernelGGL((aq_k_shard_rows<uint32_t>), dim3(grid), dim3(256), 0, 0, s->idx32, (size_t)upto, (size_t)t0, (size_t)take, p, d);
    else hipLaunchKernelGGL((aq_k_shard_rows<uint64_t>), dim3(grid), dim3(256), 0, 0, s->idx64, (size_t)upto, (size_t)t0, (size_t)take, p, d);
  }
  hipError_t e = hipMemcpy(rs_host, d, (size_t)p * sizeof(int64_t), hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, std::string("aq_shard_rows: ") + hipGetErrorString(e));
  return AQ_OK;
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
