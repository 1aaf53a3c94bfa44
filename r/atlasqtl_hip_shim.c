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
 *   atlasqtl_hip_prepare                                  prepare_data_ on the device (R/prepare_atlasqtl.R:8-87), X numeric or raw
 *       dosages; its result feeds atlasqtl_hip_vb_run, which also takes scheme, df, device-drawn initial values and n_gpus
 *       (aq_vb_run_multi: the traits sharded over the GPUs of the node inside the library).
 *
 * R API used: REAL, INTEGER, LOGICAL, RAW, TYPEOF, asReal, asInteger, asLogical, isNull, LENGTH, nrows, ncols, VECTOR_ELT,
 * STRING_ELT, CHAR, getAttrib, allocVector, allocMatrix, PROTECT/UNPROTECT, mkNamed, SET_VECTOR_ELT, ScalarReal, ScalarInteger,
 * ScalarLogical, R_MakeExternalPtr, R_ExternalPtrAddr, R_ClearExternalPtr, R_RegisterCFinalizerEx, error,
 * R_registerRoutines, R_useDynamicSymbols, R_alloc.
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

/* ---- device-side prepare_data_ (R/prepare_atlasqtl.R:8-87): .Call("atlasqtl_hip_prepare", Y, X, device) -----------------
 * X: numeric n x p matrix, or a raw n x p matrix of 0/1/2 dosages (one byte per genotype: the fp64 genotype matrix then never
 * exists in R).  Returns list(prep = <external pointer>, p_kept, bool_cst_x, bool_coll_x): the standardised X and the centred
 * Y stay on the GPU; pass `prep` as the X argument of atlasqtl_hip_vb_run.  The pointer frees the device memory when R
 * collects it. */
static void prep_finalizer(SEXP ptr) {
  aq_prep_handle h = (aq_prep_handle)R_ExternalPtrAddr(ptr);
  if (h) { aq_prep_destroy(h); R_ClearExternalPtr(ptr); }
}

SEXP atlasqtl_hip_prepare(SEXP Y, SEXP X, SEXP device) {
  aq_prep_input in;
  memset(&in, 0, sizeof(in));
  in.n = nrows(X); in.p = ncols(X); in.q = ncols(Y);
  if (nrows(Y) != in.n) error("X and Y must have the same number of samples.");
  if (TYPEOF(X) == RAWSXP) in.X_i8 = (const int8_t *)RAW(X); else in.X = REAL(X);
  in.Y = REAL(Y);
  in.device = asInteger(device);
  aq_prep_handle h = NULL;
  if (aq_prepare_data(&in, &h) != AQ_OK) error("atlasqtl (HIP) prepare_data_: %s", aq_last_error());
  SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ptr, prep_finalizer, TRUE);
  int32_t p_kept = 0;
  uint8_t *cst = (uint8_t *)R_alloc((size_t)in.p, 1), *coll = (uint8_t *)R_alloc((size_t)in.p, 1);
  if (aq_prep_info(h, &p_kept, cst, coll, NULL, NULL, NULL) != AQ_OK) { UNPROTECT(1); error("atlasqtl (HIP): %s", aq_last_error()); }
  SEXP bc = PROTECT(allocVector(LGLSXP, in.p)), bl = PROTECT(allocVector(LGLSXP, in.p));
  for (int j = 0; j < in.p; j++) { LOGICAL(bc)[j] = cst[j]; LOGICAL(bl)[j] = coll[j]; }
  const char *nm[] = {"prep", "p_kept", "n", "q", "bool_cst_x", "bool_coll_x", ""};
  SEXP out = PROTECT(mkNamed(VECSXP, nm));
  SET_VECTOR_ELT(out, 0, ptr); SET_VECTOR_ELT(out, 1, ScalarInteger(p_kept)); SET_VECTOR_ELT(out, 2, ScalarInteger(in.n));
  SET_VECTOR_ELT(out, 3, ScalarInteger(in.q)); SET_VECTOR_ELT(out, 4, bc); SET_VECTOR_ELT(out, 5, bl);
  UNPROTECT(4);
  return out;
}

/* .Call("atlasqtl_hip_vb_run", Y, X, list_hyper, list_init, anneal (numeric(3) or NULL), tol, maxit, thinned_elbo_eval, debug,
 *       device, scheme (0 global-local / 1 global-only core), df, n_gpus, init_gen (NULL, or c(seed, gam_mean, gam_sd): draw the
 *       p x q initial gam_vb / mu_beta_vb on the device -- list_init then needs no gam_vb / mu_beta_vb))
 *   ->  list(beta_vb, gam_vb, theta_vb, zeta_vb, converged, it, lb_opt, diff_lb, lb_trace_it, lb_trace)
 * X: the standardised numeric matrix, or the `prep` pointer of atlasqtl_hip_prepare (then Y is the prepared one as well; with
 * n_gpus > 1 the prepared matrices are copied back once and handed to every GPU).  n_gpus > 1 shards the traits inside the
 * library (aq_vb_run_multi: host threads + RCCL): R stays a single-threaded caller as in R/atlasqtl.R:274-278.
 * Inputs are only read (no aliasing of the caller's list_init, unlike the reference: R/atlasqtl.R:314). */
static double *opt_real(SEXP list, const char *name) {
  SEXP names = getAttrib(list, R_NamesSymbol);
  for (int i = 0; i < LENGTH(list); i++)
    if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return isNull(VECTOR_ELT(list, i)) ? NULL : REAL(VECTOR_ELT(list, i));
  return NULL;
}

SEXP atlasqtl_hip_vb_run(SEXP Y, SEXP X, SEXP list_hyper, SEXP list_init, SEXP anneal, SEXP tol, SEXP maxit,
                         SEXP thinned, SEXP debug, SEXP device, SEXP scheme, SEXP df, SEXP n_gpus, SEXP init_gen) {
  aq_vb_problem pr;
  memset(&pr, 0, sizeof(pr));
  const int ngpu = asInteger(n_gpus);
  if (ngpu < 1) error("n_gpus must be a positive integer");
  aq_prep_handle prep = TYPEOF(X) == EXTPTRSXP ? (aq_prep_handle)R_ExternalPtrAddr(X) : NULL;
  if (TYPEOF(X) == EXTPTRSXP && !prep) error("the prepared data have been released");
  if (prep) {
    int32_t p_kept = 0;
    if (aq_prep_info(prep, &p_kept, NULL, NULL, NULL, NULL, NULL) != AQ_OK) error("atlasqtl (HIP): %s", aq_last_error());
    pr.n = nrows(Y); pr.p = p_kept; pr.q = ncols(Y);
    if (ngpu == 1) {                                   /* stay on the device */
      pr.X = aq_prep_x_device(prep); pr.Y = aq_prep_y_device(prep); pr.xy_on_device = 3;
    } else {                                           /* one copy back, then to every GPU */
      double *Xh = (double *)R_alloc((size_t)pr.n * pr.p, sizeof(double)), *Yh = (double *)R_alloc((size_t)pr.n * pr.q, sizeof(double));
      if (aq_prep_get(prep, Xh, Yh) != AQ_OK) error("atlasqtl (HIP): %s", aq_last_error());
      pr.X = Xh; pr.Y = Yh;
    }
  } else {
    pr.n = nrows(X); pr.p = ncols(X); pr.q = ncols(Y);
    pr.X = REAL(X); pr.Y = REAL(Y);                    /* NA_real_ is a NaN: handled as missing */
  }
  pr.q_total = pr.q;
  pr.A2_inv = asReal(get_elt(list_hyper, "A2_inv")); pr.m0 = asReal(get_elt(list_hyper, "m0"));
  pr.nu = asReal(get_elt(list_hyper, "nu")); pr.rho = asReal(get_elt(list_hyper, "rho"));
  pr.t02 = asReal(get_elt(list_hyper, "t02"));
  pr.eta = REAL(get_elt(list_hyper, "eta")); pr.kappa = REAL(get_elt(list_hyper, "kappa"));
  pr.n0 = REAL(get_elt(list_hyper, "n0"));
  if (isNull(init_gen)) {
    pr.gam_vb = REAL(get_elt(list_init, "gam_vb")); pr.mu_beta_vb = REAL(get_elt(list_init, "mu_beta_vb"));
  } else {                                             /* auto_set_init_'s p x q draws on the device, R/set_hyper_init.R:385-387 */
    pr.init_generate = 1;
    pr.init_seed = (uint64_t)REAL(init_gen)[0]; pr.init_gam_mean = REAL(init_gen)[1]; pr.init_gam_sd = REAL(init_gen)[2];
  }
  pr.sig02_inv_vb = asReal(get_elt(list_init, "sig02_inv_vb"));
  pr.sig2_beta_vb = REAL(get_elt(list_init, "sig2_beta_vb"));
  pr.scheme = asInteger(scheme); pr.df = asInteger(df);
  pr.sig2_theta_vb = opt_real(list_init, "sig2_theta_vb");          /* not used by the global-only core */
  if (!pr.sig2_theta_vb) {
    if (pr.scheme == 0) error("list_init$sig2_theta_vb is required by the global-local core");
    double *ones = (double *)R_alloc((size_t)pr.p, sizeof(double));
    for (int j = 0; j < pr.p; j++) ones[j] = 1.0;
    pr.sig2_theta_vb = ones;
  }
  pr.tau_vb = REAL(get_elt(list_init, "tau_vb")); pr.theta_vb = REAL(get_elt(list_init, "theta_vb"));
  pr.zeta_vb = REAL(get_elt(list_init, "zeta_vb"));
  pr.has_anneal = !isNull(anneal);
  if (pr.has_anneal) for (int i = 0; i < 3; i++) pr.anneal[i] = REAL(anneal)[i];
  pr.tol = asReal(tol); pr.maxit = asInteger(maxit); pr.thinned_elbo_eval = asLogical(thinned);
  pr.debug = asLogical(debug); pr.device = asInteger(device); pr.world_size = 1;

  SEXP beta = PROTECT(allocMatrix(REALSXP, pr.p, pr.q)), gam = PROTECT(allocMatrix(REALSXP, pr.p, pr.q));
  SEXP theta = PROTECT(allocVector(REALSXP, pr.p)), zeta = PROTECT(allocVector(REALSXP, pr.q));
  const int cap = pr.maxit + 2;
  int32_t *tr_it = (int32_t *)R_alloc((size_t)cap, sizeof(int32_t));
  double *tr_lb = (double *)R_alloc((size_t)cap, sizeof(double));
  /* one entry for any number of GPUs: n_gpus = 1 is the same code path (one host thread, a one-rank communicator) */
  aq_vb_multi_out mo;
  memset(&mo, 0, sizeof(mo));
  mo.beta_vb = REAL(beta); mo.gam_vb = REAL(gam); mo.theta_vb = REAL(theta); mo.zeta_vb = REAL(zeta);
  mo.elbo_it = tr_it; mo.elbo_lb = tr_lb; mo.elbo_cap = cap;
  int32_t *devs = (int32_t *)R_alloc((size_t)ngpu, sizeof(int32_t));
  for (int r = 0; r < ngpu; r++) devs[r] = pr.device + r;           /* GPUs device .. device + n_gpus - 1 */
  if (aq_vb_run_multi(&pr, ngpu, devs, 0, &mo) != AQ_OK) {         /* incl. "ELBO not increasing monotonically. Exit." */
    UNPROTECT(4);
    error("%s", aq_last_error());
  }
  const int ntr = mo.n_elbo < cap ? mo.n_elbo : cap;
  SEXP tit = PROTECT(allocVector(INTSXP, ntr)), tlb = PROTECT(allocVector(REALSXP, ntr));
  for (int i = 0; i < ntr; i++) { INTEGER(tit)[i] = tr_it[i]; REAL(tlb)[i] = tr_lb[i]; }
  const char *nm[] = {"beta_vb", "gam_vb", "theta_vb", "zeta_vb", "converged", "it", "lb_opt", "diff_lb", "lb_trace_it", "lb_trace", ""};
  SEXP out = PROTECT(mkNamed(VECSXP, nm));
  SET_VECTOR_ELT(out, 0, beta); SET_VECTOR_ELT(out, 1, gam); SET_VECTOR_ELT(out, 2, theta); SET_VECTOR_ELT(out, 3, zeta);
  SET_VECTOR_ELT(out, 4, ScalarLogical(mo.converged)); SET_VECTOR_ELT(out, 5, ScalarInteger(mo.it));
  SET_VECTOR_ELT(out, 6, ScalarReal(mo.lb_opt)); SET_VECTOR_ELT(out, 7, ScalarReal(mo.diff_lb));
  SET_VECTOR_ELT(out, 8, tit); SET_VECTOR_ELT(out, 9, tlb);
  UNPROTECT(7);
  return out;
}

static const R_CallMethodDef CallEntries[] = {
    {"_atlasqtl_coreDualLoop", (DL_FUNC)&_atlasqtl_coreDualLoop, 15},
    {"_atlasqtl_coreDualMisLoop", (DL_FUNC)&_atlasqtl_coreDualMisLoop, 16},
    {"atlasqtl_hip_prepare", (DL_FUNC)&atlasqtl_hip_prepare, 3},
    {"atlasqtl_hip_vb_run", (DL_FUNC)&atlasqtl_hip_vb_run, 14},
    {NULL, NULL, 0}};

void R_init_atlasqtl(DllInfo *dll) {     /* replaces src/RcppExports.cpp:71-74 */
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
