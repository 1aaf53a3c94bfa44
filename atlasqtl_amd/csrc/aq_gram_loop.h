// aq_gram_loop.h -- operator-level drop-in for the reference's two native functions in
// their own (Gram-space) formulation and visiting order:
//     coreDualLoop     src/coreLoop.cpp:38-86
//     coreDualMisLoop  src/coreLoop.cpp:91-138
// One workgroup per entry of sample_q (traits are independent, src/coreLoop.cpp:58-59);
// inside a trait the j loop is the reference's sequential recursion: every thread evaluates
// the scalar update (identical inputs -> identical bits), then the p-long AXPY (:81 / :132)
// is spread over the workgroup.  FMA contraction is disabled so that each product/sum rounds
// as in the reference build (R's default flags, src/Makevars:11-12 leaves -march=native off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AqGramArgs {
  const double *cp_X;          // p x p
  const double *const *cp_X_rm;  // device array of q device pointers (MIS) or NULL
  const double *cp_Y_X;        // q x p
  double *gam_vb;
  const double *log_Phi, *log_1mPhi;
  double log_sig2_inv_vb;
  const double *log_tau_vb;
  double *m1_beta, *cp_betaX_X, *mu_beta_vb;
  const double *sig2_beta_vb;  // q (complete) or p x q (MIS)
  const double *tau_vb;
  const int32_t *shuffled_ind;
  int32_t n_ind;
  const int32_t *sample_q;
  int32_t n_q;
  double c;
  int32_t p, q;
};

__device__ __forceinline__ double aq_log1pexp_ref(double x) {   // src/coreLoop.cpp:28-33
  double m = x;
  if (x < 0) m = 0;
  return log(exp(x - m) + exp(-m)) + m;
}

template <bool MIS>
__global__ __launch_bounds__(256) void aq_gram_loop_kernel(const AqGramArgs a) {
#pragma clang fp contract(off)
  const int p = a.p, q = a.q;
  for (int ai = blockIdx.x; ai < a.n_q; ai += gridDim.x) {
    const int k = a.sample_q[ai];
    double *bx = a.cp_betaX_X + (size_t)p * k;
    const double *rm = MIS ? a.cp_X_rm[k] : nullptr;
    const double tau = a.tau_vb[k];
    double cst;
    double s2k = 0.0;
    if (MIS) {
      cst = -(a.log_tau_vb[k] + a.log_sig2_inv_vb) / 2;                                  // :108
    } else {
      s2k = a.sig2_beta_vb[k];
      cst = -(a.log_tau_vb[k] + a.log_sig2_inv_vb + log(s2k)) / 2;                       // :56
    }
    for (int b = 0; b < a.n_ind; b++) {
      const int j = a.shuffled_ind[b];
      const size_t jk = (size_t)j + (size_t)p * k;
      const size_t jj = (size_t)j + (size_t)p * j;
      const double m1_old = a.m1_beta[jk];                                               // :69
      double djj = a.cp_X[jj];
      if (MIS) djj = djj - rm[jj];
      const double r = bx[j] - m1_old * djj;                                             // :71 / :121
      const double s2 = MIS ? a.sig2_beta_vb[jk] : s2k;
      const double mu = a.c * s2 * tau * (a.cp_Y_X[(size_t)k + (size_t)q * j] - r);      // :73 / :125
      double arg;
      if (MIS)
        arg = a.c * (a.log_1mPhi[jk] - a.log_Phi[jk] - mu * mu / (2 * s2) - log(s2) / 2 + cst);   // :127-129
      else
        arg = a.c * (a.log_1mPhi[jk] - a.log_Phi[jk] - mu * mu / (2 * s2) + cst);                 // :75-77
      const double g = exp(-aq_log1pexp_ref(arg));
      const double m1 = g * mu;                                                          // :79
      const double d = m1 - m1_old;
      __syncthreads();   // every thread has read bx[j] and m1_beta[jk] before anyone overwrites them
      if (threadIdx.x == 0) {
        a.mu_beta_vb[jk] = mu;
        a.gam_vb[jk] = g;
        a.m1_beta[jk] = m1;
      }
      const double *xc = a.cp_X + (size_t)p * j;
      if (MIS) {
        const double *rc = rm + (size_t)p * j;
        for (int i = threadIdx.x; i < p; i += blockDim.x) bx[i] += d * (xc[i] - rc[i]);  // :132
      } else {
        for (int i = threadIdx.x; i < p; i += blockDim.x) bx[i] += d * xc[i];            // :81
      }
      __syncthreads();   // the AXPY is complete before the next j reads bx
    }
  }
}
