/*
 * atlasqtl_hip_shim.c -- the thin .Call shim an atlasqtl maintainer adds to bind the R package
 * to libatlasqtl_hip.so.  It is compiled by R CMD SHLIB / the package's
 * src/Makevars where R exists; it cannot be built in the development image (no R headers), so it
 * is kept to the R C API subset listed below (syntax-checked with gcc -fsyntax-only against throw-away
 * declarations of exactly that subset, outside the repository).
 *
 * It provides
 *   _atlasqtl_coreDualLoop / _atlasqtl_coreDualMisLoop   same names, arities (15 / 16) and
 *       in-place semantics as the generated Rcpp glue they replace (src/RcppExports.cpp:17,41,65-74);
 *   atlasqtl_hip_vb_run                                   the device-resident replacement of the
 *       while-loop of atlasqtl_global_local_core_ (R/atlasqtl_global_local_core.R:125-386).
 *
 * R API used: REAL, INTEGER, asReal, asInteger, LENGTH, nrows, ncols, VECTOR_ELT, getListElement
 * (local helper), allocVector, allocMatrix, PROTECT/UNPROTECT, mkNamed, SET_VECTOR_ELT, ScalarReal,
 * ScalarInteger, ScalarLogical, error, R_registerRoutines, R_useDynamicSymbols, R_alloc.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "atlasqtl_hip.h"

static SEXP get_elt(SEXP list, const char *name) {
  SEXP names = getAttrib(list, R_NamesSymbol);
  for (int i = 0; i < LENGTH(list); i++)
    if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
  error("list element '%s' not found", name);
  return R_NilValue;
}

/* .Call("_atlasqtl_coreDualLoop", cp_X, cp_Y_X, gam_vb, log_Phi, log_1_min_Phi, log_sig2_inv_vb,
 *       log_tau_vb, m1_beta, cp_betaX_X, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, sample_q, c)
 * The four in/out matrices are written in place, as the reference's Eigen::Map views are
 * (src/RcppExports.cpp:22,27-29); returns R_NilValue (:36). */
SEXP _atlasqtl_coreDualLoop(SEXP cp_X, SEXP cp_Y_X, SEXP gam_vb, SEXP log_Phi, SEXP log_1mPhi, SEXP log_sig2_inv_vb,
                            SEXP log_tau_vb, SEXP m1_beta, SEXP cp_betaX_X, SEXP mu_beta_vb, SEXP sig2_beta_vb,
                            SEXP tau_vb, SEXP shuffled_ind, SEXP sample_q, SEXP c) {
  int p = nrows(gam_vb), q = ncols(gam_vb);
  int rc = aq_core_dual_loop(REAL(cp_X), REAL(cp_Y_X), REAL(gam_vb), REAL(log_Phi), REAL(log_1mPhi),
                             asReal(log_sig2_inv_vb), REAL(log_tau_vb), REAL(m1_beta), REAL(cp_betaX_X),
                             REAL(mu_beta_vb), REAL(sig2_beta_vb), REAL(tau_vb), INTEGER(shuffled_ind),
                             LENGTH(shuffled_ind), INTEGER(sample_q), LENGTH(sample_q), asReal(c), p, q);
  if (rc != AQ_OK) error("coreDualLoop: %s", aq_last_error());
  return R_NilValue;
}

SEXP _atlasqtl_coreDualMisLoop(SEXP cp_X, SEXP cp_X_rm, SEXP cp_Y_X, SEXP gam_vb, SEXP log_Phi, SEXP log_1mPhi,
                               SEXP log_sig2_inv_vb, SEXP log_tau_vb, SEXP m1_beta, SEXP cp_betaX_X, SEXP mu_beta_vb,
                               SEXP sig2_beta_vb, SEXP tau_vb, SEXP shuffled_ind, SEXP sample_q, SEXP c) {
  int p = nrows(gam_vb), q = ncols(gam_vb);
  const double **rm = (const double **)R_alloc((size_t)q, sizeof(double *));
  for (int k = 0; k < q; k++) rm[k] = REAL(VECTOR_ELT(cp_X_rm, k));       /* as<MapMat>(cp_X_rm[k]), src/coreLoop.cpp:113 */
  int rc = aq_core_dual_mis_loop(REAL(cp_X), rm, REAL(cp_Y_X), REAL(gam_vb), REAL(log_Phi), REAL(log_1mPhi),
                                 asReal(log_sig2_inv_vb), REAL(log_tau_vb), REAL(m1_beta), REAL(cp_betaX_X),
                                 REAL(mu_beta_vb), REAL(sig2_beta_vb), REAL(tau_vb), INTEGER(shuffled_ind),
                                 LENGTH(shuffled_ind), INTEGER(sample_q), LENGTH(sample_q), asReal(c), p, q);
  if (rc != AQ_OK) error("coreDualMisLoop: %s", aq_last_error());
  return R_NilValue;
}

/* .Call("atlasqtl_hip_vb_run", Y, X, list_hyper, list_init, anneal (numeric(3) or NULL), tol, maxit,
 *       thinned_elbo_eval, debug, device)  ->  list(beta_vb, gam_vb, theta_vb, zeta_vb, converged, it, lb_opt, diff_lb)
 * Inputs are only read (no aliasing of the caller's list_init, unlike the reference: R/atlasqtl.R:314). */
SEXP atlasqtl_hip_vb_run(SEXP Y, SEXP X, SEXP list_hyper, SEXP list_init, SEXP anneal, SEXP tol, SEXP maxit,
                         SEXP thinned, SEXP debug, SEXP device) {
  aq_vb_problem pr;
  memset(&pr, 0, sizeof(pr));
  pr.n = nrows(X); pr.p = ncols(X); pr.q = ncols(Y); pr.q_total = pr.q;
  pr.X = REAL(X); pr.Y = REAL(Y);                      /* NA_real_ is a NaN: handled as missing */
  pr.A2_inv = asReal(get_elt(list_hyper, "A2_inv")); pr.m0 = asReal(get_elt(list_hyper, "m0"));
  pr.nu = asReal(get_elt(list_hyper, "nu")); pr.rho = asReal(get_elt(list_hyper, "rho"));
  pr.t02 = asReal(get_elt(list_hyper, "t02"));
  pr.eta = REAL(get_elt(list_hyper, "eta")); pr.kappa = REAL(get_elt(list_hyper, "kappa"));
  pr.n0 = REAL(get_elt(list_hyper, "n0"));
  pr.gam_vb = REAL(get_elt(list_init, "gam_vb")); pr.mu_beta_vb = REAL(get_elt(list_init, "mu_beta_vb"));
  pr.sig02_inv_vb = asReal(get_elt(list_init, "sig02_inv_vb"));
  pr.sig2_beta_vb = REAL(get_elt(list_init, "sig2_beta_vb")); pr.sig2_theta_vb = REAL(get_elt(list_init, "sig2_theta_vb"));
  pr.tau_vb = REAL(get_elt(list_init, "tau_vb")); pr.theta_vb = REAL(get_elt(list_init, "theta_vb"));
  pr.zeta_vb = REAL(get_elt(list_init, "zeta_vb"));
  pr.has_anneal = !isNull(anneal);
  if (pr.has_anneal) for (int i = 0; i < 3; i++) pr.anneal[i] = REAL(anneal)[i];
  pr.tol = asReal(tol); pr.maxit = asInteger(maxit); pr.thinned_elbo_eval = asLogical(thinned);
  pr.debug = asLogical(debug); pr.device = asInteger(device); pr.world_size = 1;

  aq_vb_handle h = NULL;
  if (aq_vb_create(&pr, &h) != AQ_OK) error("atlasqtl (HIP): %s", aq_last_error());
  if (aq_vb_run(h) != AQ_OK) {                      /* incl. "ELBO not increasing monotonically. Exit." */
    aq_vb_destroy(h);
    error("%s", aq_last_error());
  }
  aq_vb_status st;
  if (aq_vb_get_status(h, &st) != AQ_OK) {          /* e.g. a bounded in-kernel wait expired: results invalid */
    aq_vb_destroy(h);
    error("atlasqtl (HIP): %s", aq_last_error());
  }
  SEXP beta = PROTECT(allocMatrix(REALSXP, pr.p, pr.q)), gam = PROTECT(allocMatrix(REALSXP, pr.p, pr.q));
  SEXP theta = PROTECT(allocVector(REALSXP, pr.p)), zeta = PROTECT(allocVector(REALSXP, pr.q));
  int rc = aq_vb_get_result(h, REAL(beta), REAL(gam), NULL, REAL(theta), REAL(zeta), NULL, NULL, NULL, NULL);
  aq_vb_destroy(h);
  if (rc != AQ_OK) { UNPROTECT(4); error("atlasqtl (HIP): %s", aq_last_error()); }
  const char *nm[] = {"beta_vb", "gam_vb", "theta_vb", "zeta_vb", "converged", "it", "lb_opt", "diff_lb", ""};
  SEXP out = PROTECT(mkNamed(VECSXP, nm));
  SET_VECTOR_ELT(out, 0, beta); SET_VECTOR_ELT(out, 1, gam); SET_VECTOR_ELT(out, 2, theta); SET_VECTOR_ELT(out, 3, zeta);
  SET_VECTOR_ELT(out, 4, ScalarLogical(st.converged)); SET_VECTOR_ELT(out, 5, ScalarInteger(st.it));
  SET_VECTOR_ELT(out, 6, ScalarReal(st.lb_opt)); SET_VECTOR_ELT(out, 7, ScalarReal(st.diff_lb));
  UNPROTECT(5);
  return out;
}

static const R_CallMethodDef CallEntries[] = {
    {"_atlasqtl_coreDualLoop", (DL_FUNC)&_atlasqtl_coreDualLoop, 15},
    {"_atlasqtl_coreDualMisLoop", (DL_FUNC)&_atlasqtl_coreDualMisLoop, 16},
    {"atlasqtl_hip_vb_run", (DL_FUNC)&atlasqtl_hip_vb_run, 10},
    {NULL, NULL, 0}};

void R_init_atlasqtl(DllInfo *dll) {     /* replaces src/RcppExports.cpp:71-74 */
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
