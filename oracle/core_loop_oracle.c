/*
 * oracle/core_loop_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64, no FMA contraction) of the reference's native
 * inner double loop, written from the reference text:
 *
 *   oracle_core_dual_loop      follows  src/coreLoop.cpp:38-86   (coreDualLoop)
 *   oracle_core_dual_mis_loop  follows  src/coreLoop.cpp:91-138  (coreDualMisLoop)
 *   oracle_log1pexp            follows  src/coreLoop.cpp:28-33   (logOnePlusExp)
 *
 * plus an n-space port of the same recursion (oracle_nspace_loop), which is the
 * formulation the HIP product path uses (residual R_k = y_k - X beta_k instead
 * of the p x p Gram matrix) and which bench.py times as the `cpu_baseline`
 * ("port").
 *
 * PARITY STATUS: **parity unpinned**.  The reference's own native file needs
 * <RcppEigen.h> (src/utils.h:4); R, Rcpp and Eigen are absent from the build
 * image and stand-in headers are not allowed, so the reference is unbuildable
 * here and no reference-produced numbers exist.  The reference's tests hold no
 * golden vectors for this function (tests/testthat/test_convergence.R:5-7 only
 * asserts `converged`).  This file is therefore pinned by (i) being a
 * line-by-line restatement, (ii) agreement with the independent pure-R
 * statement of the same update (R/atlasqtl_global_local_core.R:188-204,
 * 208-224) restated in oracle/atlasqtl_oracle.py, (iii) agreement of the
 * Gram-space and n-space forms to <= 1e-9.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file.  The product path never does.
 *
 * Layout: everything is R column-major.  A(j,k) of a p x q matrix is A[j + p*k].
 * cp_Y_X is q x p: cp_Y_X(k,j) = cp_Y_X[k + q*j].   Indices are 0-based int32.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile); the
 * reference builds with R's default flags (no -march=native, src/Makevars:11-12)
 * so no FMA contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* src/coreLoop.cpp:28-33 */
double oracle_log1pexp(double x) {
  double m = x;
  if (x < 0) m = 0;
  return log(exp(x - m) + exp(-m)) + m;
}

/* src/coreLoop.cpp:38-86.  Mutates gam_vb, m1_beta, cp_betaX_X, mu_beta_vb in place. */
void oracle_core_dual_loop(const double *cp_X,      /* p x p */
                           const double *cp_Y_X,    /* q x p */
                           double *gam_vb,          /* p x q */
                           const double *log_Phi,   /* p x q */
                           const double *log_1mPhi, /* p x q */
                           double log_sig2_inv_vb,
                           const double *log_tau_vb, /* q */
                           double *m1_beta,          /* p x q */
                           double *cp_betaX_X,       /* p x q */
                           double *mu_beta_vb,       /* p x q */
                           const double *sig2_beta_vb, /* q */
                           const double *tau_vb,       /* q */
                           const int32_t *shuffled_ind, int32_t n_ind,
                           const int32_t *sample_q, int32_t n_q,
                           double c, int32_t p, int32_t q) {
  double *cst = (double *)malloc(sizeof(double) * (size_t)q);
  for (int k = 0; k < q; k++) /* :56 */
    cst[k] = -(log_tau_vb[k] + log_sig2_inv_vb + log(sig2_beta_vb[k])) / 2;

  for (int a = 0; a < n_q; a++) {
    int k = sample_q[a];
    double *bx = cp_betaX_X + (size_t)p * k;
    for (int b = 0; b < n_ind; b++) {
      int j = shuffled_ind[b];
      size_t jk = (size_t)j + (size_t)p * k;
      double m1_old = m1_beta[jk];                                   /* :69 */
      double r = bx[j] - m1_old * cp_X[(size_t)j + (size_t)p * j];  /* :71 */
      double mu = c * sig2_beta_vb[k] * tau_vb[k] *
                  (cp_Y_X[(size_t)k + (size_t)q * j] - r);           /* :73 */
      mu_beta_vb[jk] = mu;
      double g = exp(-oracle_log1pexp(
          c * (log_1mPhi[jk] - log_Phi[jk] - mu * mu / (2 * sig2_beta_vb[k]) + cst[k]))); /* :75-77 */
      gam_vb[jk] = g;
      double m1 = g * mu;                                            /* :79 */
      m1_beta[jk] = m1;
      double d = m1 - m1_old;
      const double *xc = cp_X + (size_t)p * j;
      for (int i = 0; i < p; i++) bx[i] += d * xc[i];               /* :81 */
    }
  }
  free(cst);
}

/* src/coreLoop.cpp:91-138.  cp_X_rm is an array of q pointers to p x p matrices
 * (the R list of the reference), sig2_beta_vb is p x q. */
void oracle_core_dual_mis_loop(const double *cp_X, const double *const *cp_X_rm,
                               const double *cp_Y_X, double *gam_vb,
                               const double *log_Phi, const double *log_1mPhi,
                               double log_sig2_inv_vb, const double *log_tau_vb,
                               double *m1_beta, double *cp_betaX_X, double *mu_beta_vb,
                               const double *sig2_beta_vb, /* p x q */
                               const double *tau_vb,
                               const int32_t *shuffled_ind, int32_t n_ind,
                               const int32_t *sample_q, int32_t n_q,
                               double c, int32_t p, int32_t q) {
  for (int a = 0; a < n_q; a++) {
    int k = sample_q[a];
    double cst = -(log_tau_vb[k] + log_sig2_inv_vb) / 2; /* :108 */
    const double *rm = cp_X_rm[k];                        /* :113 */
    double *bx = cp_betaX_X + (size_t)p * k;
    for (int b = 0; b < n_ind; b++) {
      int j = shuffled_ind[b];
      size_t jk = (size_t)j + (size_t)p * k;
      size_t jj = (size_t)j + (size_t)p * j;
      double m1_old = m1_beta[jk];
      double r = bx[j] - m1_old * (cp_X[jj] - rm[jj]); /* :121 */
      double s2 = sig2_beta_vb[jk];
      double mu = c * s2 * tau_vb[k] * (cp_Y_X[(size_t)k + (size_t)q * j] - r); /* :125 */
      mu_beta_vb[jk] = mu;
      double g = exp(-oracle_log1pexp(
          c * (log_1mPhi[jk] - log_Phi[jk] - mu * mu / (2 * s2) - log(s2) / 2 + cst))); /* :127-129 */
      gam_vb[jk] = g;
      double m1 = g * mu;
      m1_beta[jk] = m1;
      double d = m1 - m1_old;
      const double *xc = cp_X + (size_t)p * j;
      const double *rc = rm + (size_t)p * j;
      for (int i = 0; i < p; i++) bx[i] += d * (xc[i] - rc[i]); /* :132 */
    }
  }
}

/*
 * n-space port of the same recursion (what the HIP path computes):
 *   cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1*cp_X(j,j)) == x_j'(y_k - X beta_k) + (x_j'x_j) m1
 * State is the residual Rres (n x q column-major) = mask .* (Y - X m1_beta).
 * mis (n x q, 1 = observed, 0 = missing) may be NULL (complete Y); then
 * sig2_beta_vb is a q-vector and xnorm (p) holds x_j'x_j; with a mask
 * sig2_beta_vb is p x q and xnorm_mis (p x q) holds sum_i x_ij^2 mis_ik
 * (R/atlasqtl_global_local_core.R:23).
 * Same argument meaning as oracle_core_dual_loop otherwise.  Natural order.
 */
void oracle_nspace_loop(const double *X, /* n x p col-major */
                        double *Rres,    /* n x q col-major, in/out */
                        const double *mis, /* n x q or NULL */
                        const double *xnorm, /* p (mis==NULL) or p x q */
                        double *gam_vb, const double *log_Phi, const double *log_1mPhi,
                        double log_sig2_inv_vb, const double *log_tau_vb,
                        double *m1_beta, double *mu_beta_vb, const double *sig2_beta_vb,
                        const double *tau_vb, double c, int32_t n, int32_t p, int32_t q,
                        int32_t k_begin, int32_t k_end) {
  (void)q;
  for (int k = k_begin; k < k_end; k++) {
    double *rk = Rres + (size_t)n * k;
    const double *mk = mis ? mis + (size_t)n * k : NULL;
    double cst = -(log_tau_vb[k] + log_sig2_inv_vb) / 2;
    for (int j = 0; j < p; j++) {
      size_t jk = (size_t)j + (size_t)p * k;
      const double *xj = X + (size_t)n * j;
      double s = 0.0;
      for (int i = 0; i < n; i++) s += xj[i] * rk[i];
      double m1_old = m1_beta[jk];
      double d2 = mk ? xnorm[jk] : xnorm[j];
      double s2 = mk ? sig2_beta_vb[jk] : sig2_beta_vb[k];
      double mu = c * s2 * tau_vb[k] * (s + m1_old * d2);
      mu_beta_vb[jk] = mu;
      double g = exp(-oracle_log1pexp(
          c * (log_1mPhi[jk] - log_Phi[jk] - mu * mu / (2 * s2) - log(s2) / 2 + cst)));
      gam_vb[jk] = g;
      double m1 = g * mu;
      m1_beta[jk] = m1;
      double d = m1 - m1_old;
      if (mk) {
        for (int i = 0; i < n; i++) rk[i] -= d * xj[i] * mk[i];
      } else {
        for (int i = 0; i < n; i++) rk[i] -= d * xj[i];
      }
    }
  }
}
