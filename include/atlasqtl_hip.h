/*
 * atlasqtl_hip.h -- C ABI of libatlasqtl_hip.so, the MI355X (gfx950) implementation of
 * atlasqtl's variational-inference hot path.  Plain C: pointers + sizes, int status
 * returns (0 = ok), thread-local message via aq_last_error().  No torch / R types.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository hruffieux/atlasqtl @ v0.1.5).  INTEGRATION.md shows the R-side
 * binding (a .Call shim) a maintainer would add.
 *
 * All matrices are R layout: column-major fp64; indices are 0-based int32.
 */
#ifndef ATLASQTL_HIP_H_
#define ATLASQTL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQ_OK 0
#define AQ_ERR_ARG 1        /* bad argument (NULL, size, range) -- where the reference would stop()        */
#define AQ_ERR_DEVICE 2     /* no gfx950 device / HIP runtime error                                        */
#define AQ_ERR_UNSUPPORTED 3
#define AQ_ERR_NUMERIC 4    /* "ELBO not increasing monotonically" (R/atlasqtl_global_local_core.R:359-360) */

/* Message of the last failing call on this thread ("" if none). */
const char *aq_last_error(void);
/* Library version string. */
const char *aq_version(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int aq_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Operator-level entries: drop-in for the two native functions of the reference.
 *
 * aq_core_dual_loop replaces
 *     SEXP _atlasqtl_coreDualLoop(SEXP x 15)                    src/RcppExports.cpp:17-38
 *     void coreDualLoop(...)                                    src/coreLoop.cpp:38-86
 * Same positional arguments (plus explicit sizes); cp_Y_X is q x p.  gam_vb, m1_beta,
 * cp_betaX_X and mu_beta_vb (each p x q) are updated IN PLACE in the caller's host buffers,
 * exactly as the reference mutates the R objects it is handed (src/RcppExports.cpp:22,27-29).
 * Host pointers in, host pointers out; the arithmetic runs on the GPU in the reference's
 * own Gram-space form and visiting order.  Indices are range-checked (AQ_ERR_ARG) where the
 * reference has undefined behaviour.
 * ---------------------------------------------------------------------------------------- */
int aq_core_dual_loop(const double *cp_X, const double *cp_Y_X, double *gam_vb,
                      const double *log_Phi_theta_plus_zeta, const double *log_1_min_Phi_theta_plus_zeta,
                      double log_sig2_inv_vb, const double *log_tau_vb, double *m1_beta, double *cp_betaX_X,
                      double *mu_beta_vb, const double *sig2_beta_vb /* q */, const double *tau_vb,
                      const int32_t *shuffled_ind, int32_t n_ind, const int32_t *sample_q, int32_t n_q, double c,
                      int32_t p, int32_t q);

/* aq_core_dual_mis_loop replaces
 *     SEXP _atlasqtl_coreDualMisLoop(SEXP x 16)                 src/RcppExports.cpp:41-62
 *     void coreDualMisLoop(...)                                 src/coreLoop.cpp:91-138
 * cp_X_rm: array of q pointers to p x p matrices (the reference's R list); sig2_beta_vb is p x q. */
int aq_core_dual_mis_loop(const double *cp_X, const double *const *cp_X_rm, const double *cp_Y_X, double *gam_vb,
                          const double *log_Phi_theta_plus_zeta, const double *log_1_min_Phi_theta_plus_zeta,
                          double log_sig2_inv_vb, const double *log_tau_vb, double *m1_beta, double *cp_betaX_X,
                          double *mu_beta_vb, const double *sig2_beta_vb /* p x q */, const double *tau_vb,
                          const int32_t *shuffled_ind, int32_t n_ind, const int32_t *sample_q, int32_t n_q,
                          double c, int32_t p, int32_t q);

/* ------------------------------------------------------------------------------------------
 * Whole-run entries: the device-resident replacement of
 *     atlasqtl_global_local_core_(Y, X, shr_fac_inv, anneal, df, tol, maxit, verbose,
 *                                 list_hyper, list_init, ...)   R/atlasqtl_global_local_core.R:8-433
 * (its `while` loop :125-386 incl. elbo_global_local_ :440-495).  X must be the standardised
 * matrix and Y the centred matrix that prepare_data_ (R/prepare_atlasqtl.R:57-83) produces.
 * ---------------------------------------------------------------------------------------- */
typedef struct aq_vb *aq_vb_handle;

typedef struct aq_vb_problem {
  int32_t n, p;
  int32_t q;            /* traits held by THIS process (columns of Y given here)                     */
  int32_t q_total;      /* traits of the whole problem (= q on one GPU); shr_fac_inv = q_total        */
  const double *X;      /* n x p, standardised, no NaN                                                */
  const double *Y;      /* n x q, centred; NaN = missing (R/atlasqtl_global_local_core.R:19-22)       */
  /* list_hyper fields (R/set_hyper_init.R:133-134) */
  double A2_inv, m0, nu, rho, t02;
  const double *eta;    /* q */
  const double *kappa;  /* q */
  const double *n0;     /* q */
  /* list_init fields (R/set_hyper_init.R:344-346) */
  const double *gam_vb;        /* p x q */
  const double *mu_beta_vb;    /* p x q */
  double sig02_inv_vb;
  const double *sig2_beta_vb;  /* q */
  const double *sig2_theta_vb; /* p */
  const double *tau_vb;        /* q */
  const double *theta_vb;      /* p */
  const double *zeta_vb;       /* q */
  /* control (R/atlasqtl.R:179-184) */
  int32_t has_anneal;          /* 0 = anneal NULL */
  double anneal[3];            /* type (1 geometric, 2 harmonic, 3 linear), initial temperature, ladder size */
  double tol;
  int32_t maxit;
  int32_t thinned_elbo_eval;
  int32_t debug;               /* 1: non-monotone ELBO -> AQ_ERR_NUMERIC, as the reference's stop()  */
  /* placement */
  int32_t device;              /* HIP device ordinal */
  int32_t world_size;          /* number of cooperating processes (q-sharding); 1 = single GPU       */
  double *ext_reduce_main;     /* optional DEVICE buffer of aq_vb_reduce_len() doubles owned by the
                                  caller (e.g. a torch tensor) used as the all-reduce payload; NULL =
                                  the library allocates it                                            */
  double *ext_reduce_elbo;     /* optional DEVICE buffer of 8 doubles, same purpose                   */
  int32_t init_on_device;      /* 1: gam_vb and mu_beta_vb are DEVICE pointers (p x q column-major) on
                                  `device`; avoids staging 2 x 8pq bytes through the host              */
  /* SURVEY 8f N1 -- the p x q initial values of auto_set_init_ (R/set_hyper_init.R:385-387) drawn ON the
   * device: gam_vb = pnorm(N(init_gam_mean, sd = init_gam_sd)), mu_beta_vb = N(0, 1), from Philox4x32-10 keyed by
   * init_seed with counter (SNP index, trait_offset + local trait index): the draws of a trait do not depend
   * on how the traits are sharded.  gam_vb and mu_beta_vb may then be NULL.                                  */
  int32_t init_generate;
  int32_t trait_offset;        /* global index of this process's first trait (0 on one GPU)              */
  uint64_t init_seed;
  double init_gam_mean, init_gam_sd;
  int32_t xy_on_device;        /* bit 0: X is a DEVICE pointer on `device` (e.g. aq_prep_x_device of aq_prepare_data), taken as
                                  standardised and NaN-free without a host pass; bit 1: Y is a DEVICE pointer (aq_prep_y_device) */
  /* SURVEY 8f N4 -- the other drivers over the same sweep:
   *   scheme 0  atlasqtl_global_local_core_ (horseshoe, R/atlasqtl_global_local_core.R); df = 0 or 1: half-Cauchy local scales;
   *             df = 3 (R/atlasqtl_global_local_core.R:258, R/elbo.R:95-105) and df = 5, 7 (compute_integral_hs_, R/utils.R:425-568;
   *             :260-272, R/elbo.R:107-124) without annealing; other df: AQ_ERR_UNSUPPORTED
   *   scheme 1  atlasqtl_global_core_ (one global scale, R/atlasqtl_global_core.R:117-320); sig2_theta_vb of list_init and
   *             A2_inv are not used there                                                                               */
  int32_t scheme;
  int32_t df;
} aq_vb_problem;

/* Length (in doubles) of the main all-reduce payload for a problem with p predictors:
 * [ rowSums(Z) (p padded to 16) , sum(gam) , sum_k tau_k colSums(m2)_k , sum(zeta) , 5 spare ]. */
int64_t aq_vb_reduce_len(int32_t p);

int aq_vb_create(const aq_vb_problem *prob, aq_vb_handle *out);
void aq_vb_destroy(aq_vb_handle h);

/* State machine for the q-sharded multi-process run.  aq_vb_advance runs device work until a
 * collective is needed or the run is over and returns one of the codes below (<0: error, see
 * aq_last_error).  On AQ_VB_NEED_ALLREDUCE_MAIN / _ELBO the caller must SUM-all-reduce the
 * corresponding device buffer (aq_vb_reduce_ptr) across processes on the same stream order
 * (the library issues all work on the legacy default stream) and call aq_vb_advance again.
 * With world_size == 1 the codes may simply be ignored (aq_vb_run does that). */
#define AQ_VB_DONE 0
#define AQ_VB_NEED_ALLREDUCE_MAIN 1
#define AQ_VB_NEED_ALLREDUCE_ELBO 2
int aq_vb_advance(aq_vb_handle h);
/* which: 0 = main payload (aq_vb_reduce_len doubles), 1 = ELBO payload (8 doubles). */
double *aq_vb_reduce_ptr(aq_vb_handle h, int32_t which);

/* Limit the number of further sweeps aq_vb_advance may start (-1 = no limit); when the budget is
 * used up aq_vb_advance returns AQ_VB_DONE although the run is not over, and a later budget resumes it.
 * bench.py uses it to time exactly K sweeps on N processes. */
int aq_vb_set_sweep_budget(aq_vb_handle h, int32_t sweeps);

/* Single-process convenience: loops aq_vb_advance until AQ_VB_DONE (world_size must be 1). */
int aq_vb_run(aq_vb_handle h);
/* Runs at most max_sweeps further sweeps (stops earlier on convergence / maxit); world_size 1.
 * Used by bench.py to time exactly K sweeps. */
int aq_vb_run_sweeps(aq_vb_handle h, int32_t max_sweeps);

/* ------------------------------------------------------------------------------------------
 * The whole run on several GPUs of one node from ONE host process (`aq_vb_run(handle, ..., n_gpus)`), for
 * hosts without torch.distributed -- the reference's host is R, which calls the core once, single-threaded
 * (R/atlasqtl.R:274-278).  `prob` describes the WHOLE problem (q == q_total, host pointers; init_generate = 1 is allowed and
 * reproduces the single-GPU draws); the library cuts the trait axis into whole 16-trait tiles (aq_vb_partition), runs one host
 * thread and one handle per GPU and SUM-all-reduces the two small payloads of the aq_vb_advance protocol itself:
 *   transport 0  RCCL over xGMI (librccl.so is loaded at run time; distinct devices),
 *   transport 1  staged through host memory in fixed rank order (no RCCL; devices may repeat -- a one-GPU box can rehearse it).
 * devices: n_gpus HIP ordinals, or NULL for 0 .. n_gpus-1.  Results are gathered into the caller's host buffers of aq_vb_multi_out
 * (p x q column-major / q- and p-vectors; any pointer may be NULL).  Errors of any rank (e.g. AQ_ERR_NUMERIC) stop all ranks.
 * ---------------------------------------------------------------------------------------- */
typedef struct aq_vb_multi_out {
  double *beta_vb, *gam_vb, *mu_beta_vb;              /* p x q */
  double *theta_vb, *zeta_vb, *lam2_inv_vb, *sig2_theta_vb, *tau_vb, *sig2_beta_vb;
  int32_t *elbo_it;                                   /* ELBO trace: up to elbo_cap (iteration, value) pairs */
  double *elbo_lb;
  int32_t elbo_cap;
  /* written by the call */
  int32_t n_elbo, it, converged;
  double lb_opt, diff_lb, sig02_inv_vb, sig2_inv_vb;
  double seconds;                                     /* wall-clock of the run incl. set-up of the handles */
  double core_ms;                                     /* device time of the core sweep kernel, max over the GPUs */
} aq_vb_multi_out;
int aq_vb_run_multi(const aq_vb_problem *prob, int32_t n_gpus, const int32_t *devices, int32_t transport, aq_vb_multi_out *out);
/* The trait range [*k0, *k1) of part `part` of `n_parts` for q traits: whole 16-trait tiles, tile counts differing by at most
 * one (the last part also takes the ragged end).  No GPU needed. */
int aq_vb_partition(int32_t q, int32_t n_parts, int32_t part, int32_t *k0, int32_t *k1);

typedef struct aq_vb_status {
  int32_t it;            /* sweeps done                                   */
  int32_t converged;
  double lb_opt;         /* last evaluated ELBO (-inf if none yet)        */
  double diff_lb;        /* |lb_opt - lb_old|                             */
  double c;              /* inverse temperature of the NEXT sweep         */
  int32_t annealing;
  int32_t n_elbo;        /* number of ELBO evaluations so far             */
  double core_ms;        /* accumulated device time of the core sweep kernel (HIP events) */
  int32_t core_launches;
  double sig02_inv_vb, sig2_inv_vb;
  int32_t lentz_iters;   /* shared Lentz iteration count of the last non-annealed sweep */
  int32_t core_kernel;   /* which core sweep kernel this handle runs: 0 look-ahead MFMA (complete Y, or Y with NA when the
                            traits' own Gram blocks fit in HBM), 2 generic wave-per-trait (VALU), 3 round-1 masked MFMA */
  int32_t split_parts;   /* launch plan of the core kernel: workgroups sharing one trait group along the samples (1 = none), */
  int32_t tiles_per_group;   /* 16-trait tiles per workgroup (1 or 2),                                                  */
  int32_t chain_segments;    /* chained SNP segments per trait group (0 = none)                                         */
} aq_vb_status;
int aq_vb_get_status(aq_vb_handle h, aq_vb_status *st);

/* The AQ_* environment variables (launch-plan overrides for tests and experiments: AQ_TT, AQ_CHAIN, AQ_LA_C, AQ_NT3, AQ_KERNEL, ...)
 * that were SET when the handle was created, as "NAME=value NAME=value"; empty when the plan is the library's own.  Returns the
 * length of the full string (buf may be NULL).  A host that inherits its environment (an R session) can check it here. */
int32_t aq_vb_get_overrides(aq_vb_handle h, char *buf, int32_t cap);

/* ELBO trace: up to cap (iteration, value) pairs in evaluation order; returns the count. */
int32_t aq_vb_get_elbo_trace(aq_vb_handle h, int32_t *it_out, double *lb_out, int32_t cap);

/* Copy results to host buffers (any pointer may be NULL to skip).  Shapes as the reference's
 * return list (R/atlasqtl_global_local_core.R:406-428): p x q column-major matrices, p / q vectors. */
int aq_vb_get_result(aq_vb_handle h, double *beta_vb, double *gam_vb, double *mu_beta_vb, double *theta_vb,
                     double *zeta_vb, double *lam2_inv_vb, double *sig2_theta_vb, double *tau_vb,
                     double *sig2_beta_vb);

/* The residual the sweep carries in n-space, mis_pat .* (Y - X beta_vb), n x q column-major, copied to the host: what
 * cp_Y_X - cp_betaX_X encodes in the reference (src/coreLoop.cpp:71,81; R/atlasqtl_global_local_core.R:42,115).  It is
 * updated incrementally for the whole run, so its distance from Y - X beta_vb recomputed from aq_vb_get_result is the
 * accumulated rounding drift (tests/test_gpu_bigp.py). */
int aq_vb_get_residual(aq_vb_handle h, double *R_out);

/* ------------------------------------------------------------------------------------------
 * Input construction on the device: the O(n p) part of
 *     prepare_data_(Y, X, ...)                                      R/prepare_atlasqtl.R:8-87
 * i.e. X <- scale(X) (:57), rm_constant_ (R/utils.R:276-302), rm_collinear_ = duplicated(mat, MARGIN = 2) (:304-343),
 * Y <- scale(Y, center = TRUE, scale = FALSE) (:83) and the two missingness guards (:39-45, same messages).
 * X is given as fp64 (n x p column-major) or as int8 dosages X_i8 (0 / 1 / 2 ..., 1 byte per genotype; used when X is NULL),
 * so that the fp64 genotype matrix need never exist on the host.  The standardised compact matrix (n x p_kept) and the
 * centred Y stay on the device; pass aq_prep_x_device / aq_prep_y_device to aq_vb_create with xy_on_device = 1 (the handle
 * must outlive that call only).
 *   aq_prep_info   p_kept; bool_cst[p] (constant columns); bool_coll[p] (later copies of an identical column, in the
 *                  ORIGINAL numbering); dup_of[p] (original index of the kept column a removed copy equals, else -1);
 *                  the column means and n-1 standard deviations used.  Any pointer may be NULL.
 *   aq_prep_get    copies the standardised X (n x p_kept) and / or the centred Y (n x q) to the host.
 * ---------------------------------------------------------------------------------------- */
typedef struct aq_prep *aq_prep_handle;
typedef struct aq_prep_input {
  int32_t n, p, q;
  const double *X;      /* n x p fp64, or NULL */
  const int8_t *X_i8;   /* n x p int8 dosages, read when X is NULL */
  const double *Y;      /* n x q, NaN = missing */
  int32_t device;
} aq_prep_input;
int aq_prepare_data(const aq_prep_input *in, aq_prep_handle *out);
int aq_prep_info(aq_prep_handle h, int32_t *p_kept, uint8_t *bool_cst, uint8_t *bool_coll, int32_t *dup_of, double *x_mean,
                 double *x_sd);
const double *aq_prep_x_device(aq_prep_handle h);
const double *aq_prep_y_device(aq_prep_handle h);
int aq_prep_get(aq_prep_handle h, double *X_out, double *Y_out);
void aq_prep_destroy(aq_prep_handle h);

/* ------------------------------------------------------------------------------------------
 * Post-processing of the posterior inclusion probabilities on the device.
 *   aq_assign_bfdr       assign_bFDR, R/summarise_output.R:207-223: Bayesian FDR estimate of every entry of
 *                        mat_ppi (len = p*q, any layout: the function works on as.vector(mat_ppi)); ties keep their
 *                        original order as with R's order(decreasing = TRUE).  Host pointers.  Any length that fits the
 *                        device (64-bit positions from 2^32 entries on).
 *   aq_hotspot_sizes     rowSums(gam_vb > thres) / rowSums(assign_bFDR(gam_vb) < thres) and their total
 *                        (summary.atlasqtl / plot.atlasqtl, R/summarise_output.R:98-105,177-182); mat_ppi p x q
 *                        column-major on the host.
 *   aq_vb_hotspot_sizes  the same from the gam_vb resident in a handle, without copying it to the host.
 * ---------------------------------------------------------------------------------------- */
int aq_assign_bfdr(const double *mat_ppi, double *mat_fdr, int64_t len, int32_t device);
int aq_hotspot_sizes(const double *mat_ppi, int32_t p, int32_t q, double thres, int32_t fdr_adjust,
                     int64_t *rs_thres, int64_t *nb_pairwise, int32_t device);
int aq_vb_hotspot_sizes(aq_vb_handle h, double thres, int32_t fdr_adjust, int64_t *rs_thres, int64_t *nb_pairwise);
/* The same FDR thresholding when the traits are spread over several processes: assign_bFDR ranks ALL p q PPIs, so a
 * shard cannot do it alone.  Each process sorts its shard once (aq_vb_bfdr_begin) and answers, for a PPI value c,
 *   out5 = { #{ppi >= c}, sum(1 - ppi : ppi >= c), #{ppi > c}, sum(1 - ppi : ppi > c), largest ppi < c (or -1) }
 * (aq_vb_bfdr_query).  Summed over the processes these give the running mean of 1 - PPI at the end of c's tie block -- the
 * estimated FDR there, non-decreasing along the order -- so the caller bisects over c with one small all-reduce per
 * step.  aq_vb_bfdr_rows then counts, per predictor, this shard's first `upto` entries of the order plus `take` entries of
 * the tie block starting at sorted position tie_first (ties go in original order, R's order(decreasing = TRUE)).
 * atlasqtl_amd/core.py::VbRun.hotspot_sizes drives it over torch.distributed. */
int aq_vb_bfdr_begin(aq_vb_handle h);
int aq_vb_bfdr_query(aq_vb_handle h, double c, double *out5);
int aq_vb_bfdr_rows(aq_vb_handle h, int64_t upto, int64_t tie_first, int64_t take, int64_t *rs);
void aq_vb_bfdr_end(aq_vb_handle h);

/* ------------------------------------------------------------------------------------------
 * Checkpoint / resume.  The reference's checkpoint_ (R/utils.R:571-611, called at
 * R/atlasqtl_global_local_core.R:379) only writes outputs every 100 iterations and cannot resume; these
 * entries capture and restore the COMPLETE loop state between two sweeps (valid after aq_vb_run /
 * aq_vb_run_sweeps returned, or between aq_vb_advance calls that returned AQ_VB_DONE), so that a
 * restored handle continues bit-identically.  The handle receiving the state must have been created
 * for the same X, Y, hyper-parameters and device geometry (any list_init): shapes are checked.
 * ---------------------------------------------------------------------------------------- */
int64_t aq_vb_state_bytes(aq_vb_handle h);
int aq_vb_get_state(aq_vb_handle h, void *buf, int64_t cap);
int aq_vb_set_state(aq_vb_handle h, const void *buf, int64_t len);

/* ------------------------------------------------------------------------------------------
 * Test hooks for the fp64 special functions the path uses (host evaluation of the same
 * header the kernels compile): which = 0 log_ndtr, 1 digamma, 2 expint_E1 (x<=1),
 * 3 gamma_inc_upper(a=x2, x), 4 sigmoid_neg, 5 / 6 log Phi / log(1-Phi) and 7 / 8 the inverse Mills
 * ratios phi/Phi, -phi/(1-Phi) (R/utils.R:172-191) from aq_probit_terms, 9 erfcx(x), x >= 0,
 * 10 / 11 / 12 log(1-Phi) - log Phi and the two Mills ratios from the pre-pass form aq_probit_A_imr.
 * 13 the short-dependency-chain sigmoid of the SNP recursion (aq_sigmoid_neg_fast).
 * 14 - 17 compute_integral_hs_ (x = L, x2 = Q(L)); 18 / 19 / 20 log(1-Phi) - log Phi and the two Mills ratios from the piecewise
 * polynomial tables the sweep kernel evaluates (aq_probit_tab.h; tail series beyond |x| = 12); 21 / 22 / 23
 * update_annealed_lam2_inv_vb_(x = L_vb, x2 = c, df = 3 / 5 / 7) (R/update_vb.R:76-81, Kummer's 1F1).
 * 24 / 25 log Phi / log(1 - Phi) from the tables, as the ELBO pass (R/elbo.R:10-34) evaluates them (aq_log_ndtr_pair_tab).
 * Evaluates elementwise into out.  aq_special_eval_device runs the same switch in a kernel on `device`
 * (host pointers in and out): the device build of these functions (ocml, v_rcp_f64) is what the sweep executes.
 * ---------------------------------------------------------------------------------------- */
int aq_special_eval(int32_t which, const double *x, const double *x2, double *out, int64_t len);
int aq_special_eval_device(int32_t which, const double *x, const double *x2, double *out, int64_t len, int32_t device);
/* Test hook: raises the device-side flag that a bounded in-kernel wait sets when it expires (a chained SNP segment
 * waiting for its predecessor, a sample part waiting for its partners); every later aq_vb_run / aq_vb_get_status /
 * aq_vb_get_state / aq_vb_get_result on the handle must then fail with AQ_ERR_DEVICE. */
int aq_vb_debug_raise_errflag(aq_vb_handle h);
/* exp(x) E1(x) for a vector with the reference's shared Lentz stopping rule (R/utils.R:380-423);
 * host evaluation; writes the shared iteration count to *iters. */
int aq_q_approx_vec(const double *x, double *out, int64_t len, int32_t *iters);

#ifdef __cplusplus
}
#endif
#endif /* ATLASQTL_HIP_H_ */
