// aq_prepare.hip -- the O(n p) input construction of prepare_data_ on the device:
//   X <- scale(X)                                  R/prepare_atlasqtl.R:57   (centre, divide by the n - 1 standard deviation)
//   rm_constant_   (0/0 = NaN columns)             R/utils.R:276-302
//   rm_collinear_  (duplicated(mat, MARGIN = 2))   R/utils.R:304-343        (later copies of an identical column go)
//   Y <- scale(Y, center = TRUE, scale = FALSE)    R/prepare_atlasqtl.R:83   (column means over the observed entries)
//   and the two missingness guards                 R/prepare_atlasqtl.R:39-45
// X arrives as fp64 or as int8 dosages (0 / 1 / 2 ...: 1 byte per genotype, "block-standardised genotype X" of BASELINE.json
// configs[4]): at C5 that is 1 GB over PCIe instead of 8 GB, and the fp64 matrix never exists on the host.  The result stays
// on the device (compact n x p_kept fp64 + centred Y) and is handed to aq_vb_create as device pointers.
// All of it is HBM-bound streaming: one read of X for the column statistics, one read + one write for the standardised
// matrix with its column hashes, one gather pass for the compaction.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/atlasqtl_hip.h"

int aq_fail_ext(int code, const std::string &msg);   // atlasqtl_hip.hip

struct aq_prep {
  int n = 0, p = 0, q = 0, p_kept = 0, device = 0;
  double *Xs = nullptr;   // n x p_kept, standardised, compact
  double *Yc = nullptr;   // n x q, centred (NaN = missing)
  std::vector<uint8_t> bool_cst, bool_coll;   // p each (bool_coll in the ORIGINAL column numbering)
  std::vector<int32_t> dup_of;                // original index of the kept column a removed duplicate equals, else -1
  std::vector<double> mean, sd;               // p each
};

#define AQR_HIP(call)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) { rc = aq_fail_ext(AQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); goto done; } \
  } while (0)

template <typename T>
__device__ __forceinline__ double aq_xval(const T *X, size_t i) { return (double)X[i]; }

// block-wide sum in a fixed order (tree over 256 threads): deterministic, identical for identical columns
__device__ __forceinline__ double aq_block_sum(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// one workgroup per column: mean (sum / n, then one refinement pass as R's long-double colMeans would give), the n - 1
// standard deviation of the centred values, and whether the column is constant
template <typename T>
__global__ __launch_bounds__(256) void aq_k_col_stats(const T *__restrict__ X, int n, double *__restrict__ mean,
                                                     double *__restrict__ sd, uint8_t *__restrict__ cst) {
  __shared__ double sh[256];
  __shared__ int ne;
  const size_t base = (size_t)blockIdx.x * n;
  const double x0 = aq_xval(X, base);
  if (threadIdx.x == 0) ne = 0;
  __syncthreads();
  double s = 0.0;
  int diff = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = aq_xval(X, base + i);
    s += v;
    diff |= (v != x0);
  }
  if (diff) ne = 1;
  double m = aq_block_sum(s, sh) / n;
  double s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s2 += aq_xval(X, base + i) - m;
  m += aq_block_sum(s2, sh) / n;
  const bool is_cst = (ne == 0);
  if (is_cst) m = x0;                       // R centres a constant column to exactly 0 -> 0/0 = NaN -> rm_constant_
  double ss = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double d = aq_xval(X, base + i) - m;
    ss += d * d;
  }
  ss = aq_block_sum(ss, sh);
  if (threadIdx.x == 0) {
    mean[blockIdx.x] = m;
    sd[blockIdx.x] = sqrt(ss / (double)(n - 1));
    cst[blockIdx.x] = is_cst ? 1 : 0;
  }
}

__device__ __forceinline__ unsigned long long aq_mix64(unsigned long long x) {   // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// standardised values (x - mean) / sd into the compact matrix (column `dst[j]`, or skipped when dst[j] < 0) and a 128-bit
// position-dependent hash of each column's bit patterns: identical columns -> identical hashes, in any summation order
template <typename T>
__global__ __launch_bounds__(256) void aq_k_standardise(const T *__restrict__ X, int n, const double *__restrict__ mean,
                                                       const double *__restrict__ sd, const int *__restrict__ dst,
                                                       double *__restrict__ Xs, unsigned long long *__restrict__ hash) {
  __shared__ unsigned long long h0[256], h1[256];
  const int j = blockIdx.x;
  const size_t base = (size_t)j * n;
  const double m = mean[j], s = sd[j];
  const int d = dst ? dst[j] : j;
  unsigned long long a = 0, b = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = (aq_xval(X, base + i) - m) / s;
    if (d >= 0 && Xs) Xs[(size_t)d * n + i] = v;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    a += aq_mix64(bits + 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1));
    b += aq_mix64((bits ^ 0xc2b2ae3d27d4eb4full) + 0x165667b19e3779f9ull * (unsigned long long)(i + 1));
  }
  h0[threadIdx.x] = a;
  h1[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { h0[threadIdx.x] += h0[threadIdx.x + st]; h1[threadIdx.x] += h1[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && hash) { hash[2 * j] = h0[0]; hash[2 * j + 1] = h1[0]; }
}

// are the standardised columns ja and jb bit-identical?  (confirms a hash match: out[pair] = number of differing entries)
template <typename T>
__global__ __launch_bounds__(256) void aq_k_cols_equal(const T *__restrict__ X, int n, const double *__restrict__ mean,
                                                      const double *__restrict__ sd, const int *__restrict__ pairs,
                                                      int *__restrict__ out) {
  const int ja = pairs[2 * blockIdx.x], jb = pairs[2 * blockIdx.x + 1];
  int diff = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double va = (aq_xval(X, (size_t)ja * n + i) - mean[ja]) / sd[ja];
    const double vb = (aq_xval(X, (size_t)jb * n + i) - mean[jb]) / sd[jb];
    diff += (__double_as_longlong(va) != __double_as_longlong(vb));
  }
  if (diff) atomicAdd(&out[blockIdx.x], diff);
}

// Y <- scale(Y, center = TRUE, scale = FALSE): column means over the observed entries; nobs per column
__global__ __launch_bounds__(256) void aq_k_centre_y(const double *__restrict__ Y, int n, double *__restrict__ Yc,
                                                    int *__restrict__ nobs) {
  __shared__ double sh[256];
  const size_t base = (size_t)blockIdx.x * n;
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = Y[base + i];
    if (v == v) { s += v; c += 1.0; }
  }
  s = aq_block_sum(s, sh);
  c = aq_block_sum(c, sh);
  double m = c > 0 ? s / c : 0.0;
  double s2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = Y[base + i];
    if (v == v) s2 += v - m;
  }
  if (c > 0) m += aq_block_sum(s2, sh) / c;
  else aq_block_sum(s2, sh);
  for (int i = threadIdx.x; i < n; i += 256) Yc[base + i] = Y[base + i] - m;   // NaN - m = NaN
  if (threadIdx.x == 0) nobs[blockIdx.x] = (int)c;
}

template <typename T>
static int aq_prepare_x(aq_prep *h, const T *X_host) {
  int rc = AQ_OK;
  const int n = h->n, p = h->p;
  const size_t np = (size_t)n * p;
  T *dX = nullptr;
  double *dmean = nullptr, *dsd = nullptr;
  uint8_t *dcst = nullptr;
  unsigned long long *dhash = nullptr;
  int *ddst = nullptr, *dpairs = nullptr, *dout = nullptr;
  std::vector<unsigned long long> hash((size_t)2 * p);
  std::vector<int> dst(p, -1);
  {
    AQR_HIP(hipMalloc((void **)&dX, np * sizeof(T)));
    AQR_HIP(hipMemcpy(dX, X_host, np * sizeof(T), hipMemcpyHostToDevice));
    AQR_HIP(hipMalloc((void **)&dmean, (size_t)p * sizeof(double)));
    AQR_HIP(hipMalloc((void **)&dsd, (size_t)p * sizeof(double)));
    AQR_HIP(hipMalloc((void **)&dcst, (size_t)p));
    AQR_HIP(hipMalloc((void **)&dhash, (size_t)2 * p * sizeof(unsigned long long)));
    hipLaunchKernelGGL((aq_k_col_stats<T>), dim3(p), dim3(256), 0, 0, dX, n, dmean, dsd, dcst);
    // first pass over the standardised values: hashes only (nothing is written yet: the compact column index needs them)
    hipLaunchKernelGGL((aq_k_standardise<T>), dim3(p), dim3(256), 0, 0, dX, n, dmean, dsd, (const int *)nullptr, (double *)nullptr, dhash);
    AQR_HIP(hipGetLastError());
    h->bool_cst.assign(p, 0); h->bool_coll.assign(p, 0); h->dup_of.assign(p, -1); h->mean.resize(p); h->sd.resize(p);
    AQR_HIP(hipMemcpy(h->bool_cst.data(), dcst, (size_t)p, hipMemcpyDeviceToHost));
    AQR_HIP(hipMemcpy(h->mean.data(), dmean, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    AQR_HIP(hipMemcpy(h->sd.data(), dsd, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
    AQR_HIP(hipMemcpy(hash.data(), dhash, hash.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (const char *e = getenv("AQ_PREP_HASH_MASK")) {   // test hook: truncate the hashes so that different columns collide
      const unsigned long long mask = strtoull(e, nullptr, 0);
      for (auto &v : hash) v &= mask;
    }
    // duplicated(mat, MARGIN = 2) among the non-constant columns: the first column of each hash class is kept; a later
    // member is removed once a bitwise comparison on the device has confirmed it (a 128-bit collision is not ruled out)
    struct Key { unsigned long long a, b; bool operator==(const Key &o) const { return a == o.a && b == o.b; } };
    struct KeyHash { size_t operator()(const Key &k) const { return (size_t)(k.a ^ (k.b * 0x9e3779b97f4a7c15ull)); } };
    std::unordered_map<Key, std::vector<int>, KeyHash> seen;   // hash -> kept columns with that hash (normally one)
    std::vector<int> pairs, cand;
    for (int j = 0; j < p; j++) {
      if (h->bool_cst[j]) continue;
      Key k{hash[2 * (size_t)j], hash[2 * (size_t)j + 1]};
      auto it = seen.find(k);
      if (it == seen.end()) { seen[k] = {j}; continue; }
      cand.push_back(j);
      pairs.push_back(it->second[0]);
      pairs.push_back(j);
    }
    if (!cand.empty()) {
      std::vector<int> out(cand.size(), 0);
      AQR_HIP(hipMalloc((void **)&dpairs, pairs.size() * sizeof(int)));
      AQR_HIP(hipMalloc((void **)&dout, out.size() * sizeof(int)));
      AQR_HIP(hipMemcpy(dpairs, pairs.data(), pairs.size() * sizeof(int), hipMemcpyHostToDevice));
      AQR_HIP(hipMemset(dout, 0, out.size() * sizeof(int)));
      hipLaunchKernelGGL((aq_k_cols_equal<T>), dim3((unsigned)cand.size()), dim3(256), 0, 0, dX, n, dmean, dsd, dpairs, dout);
      AQR_HIP(hipMemcpy(out.data(), dout, out.size() * sizeof(int), hipMemcpyDeviceToHost));
      // A candidate the bitwise comparison found DIFFERENT from the first column of its hash class (a 128-bit collision) is a
      // column of its own: it joins the class, and every later candidate of the class that differs from the first member is
      // compared with the further members as well, in column order (one pair per launch: this never happens in practice).
      auto equal_on_device = [&](int ca, int cb, bool *eq) -> int {
        const int pr2[2] = {ca, cb};
        int res = 0;
        if (hipMemcpy(dpairs, pr2, sizeof(pr2), hipMemcpyHostToDevice) != hipSuccess || hipMemset(dout, 0, sizeof(int)) != hipSuccess)
          return aq_fail_ext(AQ_ERR_DEVICE, "aq_prepare_data: duplicate check failed");
        hipLaunchKernelGGL((aq_k_cols_equal<T>), dim3(1), dim3(256), 0, 0, dX, n, dmean, dsd, dpairs, dout);
        if (hipMemcpy(&res, dout, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
          return aq_fail_ext(AQ_ERR_DEVICE, "aq_prepare_data: duplicate check failed");
        *eq = (res == 0);
        return AQ_OK;
      };
      for (size_t c = 0; c < cand.size(); c++) {
        const int j = cand[c];
        if (out[c] == 0) { h->bool_coll[j] = 1; h->dup_of[j] = pairs[2 * c]; continue; }
        std::vector<int> &members = seen[Key{hash[2 * (size_t)j], hash[2 * (size_t)j + 1]}];
        bool dup = false;
        for (size_t m = 1; m < members.size() && !dup; m++) {
          bool eq = false;
          rc = equal_on_device(members[m], j, &eq);
          if (rc != AQ_OK) goto done;
          if (eq) { h->bool_coll[j] = 1; h->dup_of[j] = members[m]; dup = true; }
        }
        if (!dup) members.push_back(j);
      }
    }
    int kept = 0;
    for (int j = 0; j < p; j++)
      if (!h->bool_cst[j] && !h->bool_coll[j]) dst[j] = kept++;
    h->p_kept = kept;
    if (kept < 1) { rc = aq_fail_ext(AQ_ERR_ARG, "There must be at least 1 non-constant candidate predictor stored in X."); goto done; }
    AQR_HIP(hipMalloc((void **)&ddst, (size_t)p * sizeof(int)));
    AQR_HIP(hipMemcpy(ddst, dst.data(), (size_t)p * sizeof(int), hipMemcpyHostToDevice));
    AQR_HIP(hipMalloc((void **)&h->Xs, (size_t)n * kept * sizeof(double)));
    hipLaunchKernelGGL((aq_k_standardise<T>), dim3(p), dim3(256), 0, 0, dX, n, dmean, dsd, ddst, h->Xs, (unsigned long long *)nullptr);
    AQR_HIP(hipGetLastError());
    AQR_HIP(hipDeviceSynchronize());
  }
done:
  if (dX) hipFree(dX);
  if (dmean) hipFree(dmean);
  if (dsd) hipFree(dsd);
  if (dcst) hipFree(dcst);
  if (dhash) hipFree(dhash);
  if (ddst) hipFree(ddst);
  if (dpairs) hipFree(dpairs);
  if (dout) hipFree(dout);
  return rc;
}

extern "C" void aq_prep_destroy(aq_prep_handle h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->Xs) hipFree(h->Xs);
  if (h->Yc) hipFree(h->Yc);
  delete h;
}

extern "C" int aq_prepare_data(const aq_prep_input *in, aq_prep_handle *out) {
  if (!in || !out) return aq_fail_ext(AQ_ERR_ARG, "aq_prepare_data: NULL argument");
  *out = nullptr;
  if (in->n < 2 || in->p < 1 || in->q < 1) return aq_fail_ext(AQ_ERR_ARG, "aq_prepare_data: n >= 2, p >= 1, q >= 1 required");
  if ((!in->X && !in->X_i8) || !in->Y) return aq_fail_ext(AQ_ERR_ARG, "aq_prepare_data: NULL data pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return aq_fail_ext(AQ_ERR_DEVICE, "no HIP device visible: libatlasqtl_hip has no CPU fallback (MI355X / gfx950 required)");
  if (in->device < 0 || in->device >= ndev) return aq_fail_ext(AQ_ERR_ARG, "device ordinal out of range");
  if (hipSetDevice(in->device) != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, "hipSetDevice failed");
  const size_t np = (size_t)in->n * in->p, nq = (size_t)in->n * in->q;
  if (in->X)
    for (size_t i = 0; i < np; i++)   // check_structure_(X, "matrix", "numeric"): no NA, finite (R/utils.R:34-100)
      if (!std::isfinite(in->X[i])) return aq_fail_ext(AQ_ERR_ARG, "X must be a non-empty a numeric matrix, finite without missing value.");
  aq_prep *h = new aq_prep();
  h->n = in->n; h->p = in->p; h->q = in->q; h->device = in->device;
  int rc = in->X ? aq_prepare_x<double>(h, in->X) : aq_prepare_x<int8_t>(h, in->X_i8);
  double *dY = nullptr;
  int *dnobs = nullptr;
  std::vector<int> nobs(in->q);
  if (rc != AQ_OK) goto done;
  AQR_HIP(hipMalloc((void **)&dY, nq * sizeof(double)));
  AQR_HIP(hipMalloc((void **)&h->Yc, nq * sizeof(double)));
  AQR_HIP(hipMalloc((void **)&dnobs, (size_t)in->q * sizeof(int)));
  AQR_HIP(hipMemcpy(dY, in->Y, nq * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(aq_k_centre_y, dim3(in->q), dim3(256), 0, 0, dY, in->n, h->Yc, dnobs);
  AQR_HIP(hipGetLastError());
  AQR_HIP(hipMemcpy(nobs.data(), dnobs, nobs.size() * sizeof(int), hipMemcpyDeviceToHost));
  {
    long long tot = 0;
    std::string low;
    for (int k = 0; k < in->q; k++) {
      tot += nobs[k];
      if ((double)nobs[k] / in->n < 0.025) low += (low.empty() ? "" : " ") + std::to_string(k + 1);
    }
    if ((double)tot / ((double)in->n * in->q) < 0.05) { rc = aq_fail_ext(AQ_ERR_ARG, "Too few non-NA values in matrix Y. Exit."); goto done; }
    if (!low.empty()) {
      rc = aq_fail_ext(AQ_ERR_ARG, "Column(s) " + low + " of matrix Y have more than 97.5% missing values, and should be removed. Exit.");
      goto done;
    }
  }
done:
  if (dY) hipFree(dY);
  if (dnobs) hipFree(dnobs);
  if (rc != AQ_OK) { aq_prep_destroy(h); return rc; }
  *out = h;
  return AQ_OK;
}

extern "C" int aq_prep_info(aq_prep_handle h, int32_t *p_kept, uint8_t *bool_cst, uint8_t *bool_coll, int32_t *dup_of,
                            double *x_mean, double *x_sd) {
  if (!h) return aq_fail_ext(AQ_ERR_ARG, "NULL handle");
  if (p_kept) *p_kept = h->p_kept;
  if (bool_cst) std::copy(h->bool_cst.begin(), h->bool_cst.end(), bool_cst);
  if (bool_coll) std::copy(h->bool_coll.begin(), h->bool_coll.end(), bool_coll);
  if (dup_of) std::copy(h->dup_of.begin(), h->dup_of.end(), dup_of);
  if (x_mean) std::copy(h->mean.begin(), h->mean.end(), x_mean);
  if (x_sd) std::copy(h->sd.begin(), h->sd.end(), x_sd);
  return AQ_OK;
}

extern "C" const double *aq_prep_x_device(aq_prep_handle h) { return h ? h->Xs : nullptr; }
extern "C" const double *aq_prep_y_device(aq_prep_handle h) { return h ? h->Yc : nullptr; }

extern "C" int aq_prep_get(aq_prep_handle h, double *X_out, double *Y_out) {
  if (!h) return aq_fail_ext(AQ_ERR_ARG, "NULL handle");
  if (hipSetDevice(h->device) != hipSuccess) return aq_fail_ext(AQ_ERR_DEVICE, "hipSetDevice failed");
  if (X_out && hipMemcpy(X_out, h->Xs, (size_t)h->n * h->p_kept * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return aq_fail_ext(AQ_ERR_DEVICE, "aq_prep_get: copy of X failed");
  if (Y_out && hipMemcpy(Y_out, h->Yc, (size_t)h->n * h->q * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return aq_fail_ext(AQ_ERR_DEVICE, "aq_prep_get: copy of Y failed");
  return AQ_OK;
}
