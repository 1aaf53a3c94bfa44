// atlasqtl_hip.hip -- host side of libatlasqtl_hip.so: the C ABI declared in
// include/atlasqtl_hip.h, the device-resident VB state and the sweep sequencing that
// replaces the reference's R-level loop (R/atlasqtl_global_local_core.R:125-386).
// gfx950 only.  No CPU fallback: every entry fails loudly without a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/atlasqtl_hip.h"
#include "aq_core_sweep.h"
#include "aq_launch_la.h"
#include "aq_gram_loop.h"
#include "aq_special.h"
#include "aq_trait_wave.h"
#include "aq_core_sweep_mis.h"
#include "aq_vec_kernels.h"

// ------------------------------------------------------------------ errors ----
static thread_local std::string g_err;
static int aq_fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
int aq_fail_ext(int code, const std::string &msg) { return aq_fail(code, msg); }   // for aq_postproc.hip
// aq_postproc.hip (hipCUB sort / scan)
int aq_bfdr_device(const double *d_ppi, double *d_fdr, int64_t len);
int aq_row_count_device(const double *d_m, int64_t *d_rs, int p, int q, double thres, int lt);
struct aq_shard_sorted;
int aq_shard_sort(const double *d_ppi, int64_t len, aq_shard_sorted **out);
int aq_shard_query(const aq_shard_sorted *s, double c, double out[5]);
int aq_shard_rows(const aq_shard_sorted *s, int64_t upto, int64_t t0, int64_t take, int p, int64_t *rs_host);
void aq_shard_free(aq_shard_sorted *s);
#define AQ_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return aq_fail(AQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                                        std::to_string(__LINE__) + ")");                               \
  } while (0)

extern "C" const char *aq_last_error(void) { return g_err.c_str(); }
extern "C" const char *aq_version(void) { return "atlasqtl_hip 0.1.0 (gfx950)"; }
extern "C" int aq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int aq_need_device(int device) {
  int n = aq_device_count();
  if (n <= 0)
    return aq_fail(AQ_ERR_DEVICE, "no HIP device visible: libatlasqtl_hip has no CPU fallback (MI355X / gfx950 required)");
  if (device < 0 || device >= n) return aq_fail(AQ_ERR_ARG, "device ordinal out of range");
  AQ_HIP(hipSetDevice(device));
  return AQ_OK;
}

// ------------------------------------------------- f64 MFMA D-layout probe ----
// D = A(16x4) * B(4x16) with A[i][k] = (k==0 ? i : 0), B[0][j] = 1  =>  D[i][j] = i.
// Reading reg 1 of lane 16 tells which row the (reg, lane>>4) pair maps to.
__global__ void aq_k_probe_dlayout(double *out) {
  int lane = threadIdx.x & 63;
  double av = ((lane >> 4) == 0) ? (double)(lane & 15) : 0.0;
  double bv = ((lane >> 4) == 0) ? 1.0 : 0.0;
  aq_d4 acc = {0, 0, 0, 0};
  acc = aq_mfma(av, bv, acc);
  for (int r = 0; r < 4; r++) out[lane * 4 + r] = acc[r];
}
static int g_dmode = -1;
static int aq_probe_dmode(int *dmode) {
  if (g_dmode >= 0) {
    *dmode = g_dmode;
    return AQ_OK;
  }
  double *d;
  AQ_HIP(hipMalloc(&d, 256 * sizeof(double)));
  hipLaunchKernelGGL(aq_k_probe_dlayout, dim3(1), dim3(64), 0, 0, d);
  double h[256];
  AQ_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  AQ_HIP(hipFree(d));
  // lane 16 (g = 1), reg 1: row = g + 4*reg = 5 (mode 0)  or  4*g + reg = 5 (mode 1)?  ambiguous -> use lane 16 reg 0
  // lane 16, reg 0: mode 0 -> row 1, mode 1 -> row 4.
  double v = h[16 * 4 + 0];
  int mode;
  if (v == 1.0) mode = 0;
  else if (v == 4.0) mode = 1;
  else return aq_fail(AQ_ERR_DEVICE, "unexpected v_mfma_f64_16x16x4 accumulator layout (probe value " + std::to_string(v) + ")");
  // full check of the chosen map
  for (int lane = 0; lane < 64; lane++)
    for (int r = 0; r < 4; r++) {
      int g = lane >> 4;
      int row = mode ? 4 * g + r : 4 * r + g;
      if (h[lane * 4 + r] != (double)row)
        return aq_fail(AQ_ERR_DEVICE, "v_mfma_f64_16x16x4 accumulator layout does not match either known map");
    }
  g_dmode = mode;
  *dmode = mode;
  return AQ_OK;
}

// ----------------------------------------------------------------- helpers ----
template <typename T>
static int aq_dalloc(T **ptr, size_t count) {
  AQ_HIP(hipMalloc((void **)ptr, count * sizeof(T)));
  AQ_HIP(hipMemset(*ptr, 0, count * sizeof(T)));
  return AQ_OK;
}
#define AQ_TRY(x)            \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != AQ_OK) return rc_; \
  } while (0)

static int aq_upload_padded(double *dst, const double *src, size_t n, size_t n_pad) {
  AQ_HIP(hipMemset(dst, 0, n_pad * sizeof(double)));
  AQ_HIP(hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
  return AQ_OK;
}

// ------------------------------------------------------------------ state ----
struct aq_vb {
  int n, p, q, q_total, p_pad, q_pad, n_pad, nb, ntile, NT, NW, dmode, device, world, trait_offset = 0;
  // hyper / control
  double A2_inv, m0, nu, rho, t02, t02_inv, shr;
  bool has_anneal;
  int scheme = 0, df = 1;   // scheme 1 = global-only core (atlasqtl_global_core_); df of the horseshoe's half-t prior (1 or 3)
  double anneal[3];
  std::vector<double> ladder;
  double tol;
  int maxit;
  bool thinned, debug;
  // device buffers
  double2 *XA = nullptr, *XU = nullptr;
  double *G = nullptr, *Gx = nullptr, *R = nullptr, *gam = nullptr, *mu = nullptr;
  double *theta = nullptr, *sig2_theta = nullptr, *L = nullptr, *lam2_inv = nullptr, *Q = nullptr, *ppart = nullptr;
  double *eta_h = nullptr, *kappa_h = nullptr, *n0 = nullptr, *nobs = nullptr;
  double *zeta = nullptr, *tau = nullptr, *sig2b = nullptr, *log_tau = nullptr, *eta_vb = nullptr, *kappa_vb = nullptr;
  double *coef = nullptr, *inv2s = nullptr, *cst = nullptr, *sums = nullptr, *rowA = nullptr, *rowGB = nullptr;
  double *Aarr = nullptr, *Barr = nullptr, *colApart = nullptr;
  bool use_la = false;   // look-ahead kernel (aq_core_sweep_la.h)
  int laC = 1;           // look-ahead kernel: workgroups (sample parts) per trait group, n > 1056
  int max_missing = 0;   // most missing samples of one trait
  bool la_mask = false;  // look-ahead kernel, MASK instances: Y with missing values, per-trait Gram blocks precomputed into GK
  double *GK = nullptr;  // [ntile][nb][AQ_GK_STRIDE]
  int TT = 1;            // look-ahead kernel: 16-trait tiles per workgroup (2 when there are enough tiles to fill the chip)
  int stagger = 0;       // look-ahead kernel: tile at which a matrix wave releases its SIMD partner into the phase (0 = off)
  int NT2 = 0;           // look-ahead kernel: tiles of matrix waves 4,5,6 (NT: waves 0,1,2)
  bool use_tw = false;   // generic wave-per-trait kernel (aq_trait_wave.h): missing Y, or n beyond the MFMA kernels
  int NE = 0;            // samples per lane of the generic kernel
  int WPT = 1;           // generic kernel: waves (and workgroups) sharing one trait (tile); also rowGB rows per tile
  int tw_ns = 4;         // generic kernel: SNP columns staged in LDS at a time
  bool use_mis = false;  // masked blocked MFMA kernel (aq_core_sweep_mis.h): missing Y, n <= 2048
  double *XR = nullptr;  // [nb][NR][16] row-major SNP panels (gather source of the per-trait Gram corrections)
  int *midx = nullptr, *mcnt4 = nullptr;   // per-trait lists of missing samples
  int Mmax = 0, NR = 0;
  int misC = 1;          // masked kernel: workgroups (sample parts) per trait tile
  double *Pbuf = nullptr, *rnpart = nullptr;
  int *pflag = nullptr;
  double *Xcm = nullptr, *mis = nullptr, *XN = nullptr;
  int ncu = 256;
  int la_xtouch = 1;        // helper waves warm the L2 with the next phase's X operand panels (AQ_XTOUCH=0 switches it off)
  bool la_nt3_pinned = false;   // AQ_NT3 given: the annealed sweeps keep the geometry as well
  int NT3x = -1;                // look-ahead kernel, two-tile instances: 9 residual tiles on the recurrence wave (geometry NT / NT / 9), -1 = aq_la_nt3
  int la_mprio = 1;         // matrix waves: hand-offs at raised priority (AQ_MPRIO=0 switches it off): C3 34.84 -> 34.67 ms, C3 + 5 % NA 49.9 -> 47.1
  int la_hprio = 0;         // s_setprio level of the helper wave (AQ_HPRIO; 1 for the unsplit MASK instances, see below)
  int la_xhelper = 0;       // sample split of the look-ahead kernel: exchange on the helper wave (long matrix phases) or on the recurrence wave
  int chain = 0;            // > 1: chained-segment launch with that many SNP segments (aq_core_sweep_la.h, SEG)
  int *done = nullptr, *errflag = nullptr;
  bool pre_done = false;
  bool fused = false;    // look-ahead kernel: the pre-pass (A, b, sums of a) is computed inside the sweep kernel
  double *red = nullptr, *ered = nullptr, *Hpart = nullptr;
  bool own_red = false, own_ered = false;
  AqScalars *sc = nullptr;
  int pblk = 0, nHchunk = 0, rows_per_chunk = 0;
  // host-side loop state (R/atlasqtl_global_local_core.R:73-97,121-123)
  bool annealing = false;
  double c = 1.0, c_s = 1.0, sig2_zeta = 0.0, vec_sum_log_det_zeta = 0.0;
  int it_init = 1;
  std::vector<double> times_conv_sched;
  std::vector<int> batch_conv_sched;
  int ind_batch_conv = 0, batch_conv = 1;
  bool converged = false;
  double lb_new = -std::numeric_limits<double>::infinity(), lb_old = -std::numeric_limits<double>::infinity();
  int it = 0;
  int phase = 0;   // 0 start, 1 after init reduce, 2 sweep part A next, 3 after main reduce, 4 after elbo reduce
  bool has_missing = false;
  std::vector<int> trace_it;
  std::vector<double> trace_lb;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double core_ms_acc = 0.0;
  int core_launches = 0;
  aq_shard_sorted *bf = nullptr;   // this rank's sorted PPIs between aq_vb_bfdr_begin and aq_vb_bfdr_end
  bool failed = false;
  int fail_code = AQ_ERR_NUMERIC;   // why the handle failed (reported again by every later advance)
  std::string fail_msg;
  bool errflag_forced = false;   // test hook aq_vb_debug_raise_errflag
  int budget = -1;
  std::string overrides;   // "NAME=value ..." of the AQ_* environment hooks that were set when the handle was created
};

// Environment overrides of the launch plan (test and experiment hooks: AQ_TT, AQ_CHAIN, AQ_LA_C, ...).  Every one that is SET when a
// handle is created is recorded in the handle and reported by aq_vb_get_overrides, so that a host which inherits such a variable
// from its environment -- an R session, a batch script -- can see that the plan is not the library's own.
static const char *aq_env(aq_vb *s, const char *name) {
  const char *v = getenv(name);
  if (v && s) {
    const std::string item = std::string(name) + "=" + v;
    if (s->overrides.find(item) == std::string::npos) s->overrides += (s->overrides.empty() ? "" : " ") + item;
  }
  return v;
}

static void aq_free_all(aq_vb *s) {
  if (!s) return;
  hipSetDevice(s->device);
  void *ptrs[] = {s->GK, s->Pbuf, s->rnpart, s->pflag, s->XR, s->midx, s->mcnt4, s->Xcm, s->mis, s->XN, s->XA, s->XU, s->G, s->Gx, s->R, s->gam, s->mu, s->theta, s->sig2_theta, s->L, s->lam2_inv, s->Q, s->ppart,
                  s->eta_h, s->kappa_h, s->n0, s->nobs, s->zeta, s->tau, s->sig2b, s->log_tau, s->eta_vb, s->kappa_vb,
                  s->coef, s->inv2s, s->cst, s->sums, s->rowA, s->rowGB, s->Aarr, s->Barr, s->colApart, s->Hpart, s->sc};
  for (void *ptr : ptrs)
    if (ptr) hipFree(ptr);
  if (s->own_red && s->red) hipFree(s->red);
  if (s->own_ered && s->ered) hipFree(s->ered);
  for (auto &e : s->ev) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  if (s->done) hipFree(s->done);
  if (s->errflag) hipFree(s->errflag);
  if (s->bf) aq_shard_free(s->bf);
  delete s;
}

extern "C" int64_t aq_vb_reduce_len(int32_t p) { return (int64_t)((p + 15) / 16) * 16 + AQ_RED_EXTRA; }

// get_annealing_ladder_, R/utils.R:108-146
static std::vector<double> aq_ladder(const double anneal[3]) {
  double k_m = 1.0 / anneal[1];
  int m = (int)std::llround(anneal[2]);
  std::vector<double> l(m);
  int type = (int)std::llround(anneal[0]);
  if (type == 1) {
    double delta = std::pow(k_m, 1.0 / (1 - m)) - 1;
    for (int i = 0; i < m; i++) l[i] = std::pow(1 + delta, 1.0 - (double)(m - i));
  } else if (type == 2) {
    double delta = (1 / k_m - 1) / (m - 1);
    for (int i = 0; i < m; i++) l[i] = 1.0 / (1 + delta * ((double)(m - i) - 1));
  } else {
    double delta = (1 - k_m) / (m - 1);
    for (int i = 0; i < m; i++) l[i] = k_m + delta * (double)i;
  }
  return l;
}

static int aq_launch_tw(aq_vb *s, int mode, double c) {
  AqTwArgs t;
  t.X = s->Xcm; t.R = s->R; t.mis = s->mis; t.XN = s->XN; t.gam = s->gam; t.mu = s->mu; t.Aarr = s->Aarr; t.Barr = s->Barr;
  t.tau = s->tau; t.log_tau = s->log_tau; t.sig2b = s->sig2b; t.sc = s->sc; t.sums = s->sums; t.rowGB = s->rowGB; t.c = c;
  t.n = s->n; t.p = s->p; t.q = s->q; t.n_pad = s->n_pad; t.p_pad = s->p_pad; t.q_pad = s->q_pad; t.ntile = s->ntile;
  t.mode = mode; t.complete = s->has_missing ? 0 : 1;
  t.ns = s->tw_ns;
  size_t lds = (size_t)(s->tw_ns * s->n_pad + 8 * 256 + 32) * sizeof(double);
#define AQ_TW(NE_, WPT_)                                                                                       \
  if (s->NE == NE_ && s->WPT == WPT_) {                                                                        \
    AQ_HIP(hipFuncSetAttribute((const void *)aq_trait_wave_kernel<NE_, WPT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((aq_trait_wave_kernel<NE_, WPT_>), dim3(s->ntile * WPT_), dim3(1024), lds, 0, t);     \
  } else
  AQ_TW(4, 1) AQ_TW(8, 1) AQ_TW(16, 1) AQ_TW(32, 1)
  AQ_TW(4, 2) AQ_TW(8, 2) AQ_TW(16, 2) AQ_TW(32, 2) AQ_TW(40, 2)
  AQ_TW(4, 4) AQ_TW(8, 4) AQ_TW(16, 4) AQ_TW(32, 4) AQ_TW(40, 4)
  { return aq_fail(AQ_ERR_UNSUPPORTED, "no generic kernel instantiation for this n"); }
#undef AQ_TW
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

static int aq_launch_mis(aq_vb *s, int mode, double c) {
  AqMisArgs t;
  t.XA = s->XA; t.XU = s->XU; t.G = s->G; t.XR = s->XR; t.R = s->R; t.mis = s->mis; t.gam = s->gam; t.mu = s->mu;
  t.Aarr = s->Aarr; t.Barr = s->Barr; t.tau = s->tau; t.log_tau = s->log_tau; t.sig2b = s->sig2b; t.sc = s->sc;
  t.midx = s->midx; t.mcnt4 = s->mcnt4; t.sums = s->sums; t.rowGB = s->rowGB; t.c = c;
  t.p = s->p; t.q = s->q; t.p_pad = s->p_pad; t.q_pad = s->q_pad; t.n_pad = s->n_pad; t.nb = s->nb; t.ntile = s->ntile;
  t.dmode = s->dmode; t.mode = mode; t.NR = s->NR; t.Mmax = s->Mmax;
  t.C = s->misC; t.Pbuf = s->Pbuf; t.pflag = s->pflag; t.errflag = s->errflag; t.rnpart = s->rnpart;
  const int nseg = (mode == 0 && s->misC == 1 && s->chain > 1) ? s->chain : 1;
  t.nseg = nseg; t.done = s->done;
  if (nseg > 1) AQ_HIP(hipMemsetAsync(s->done, 0, (size_t)s->ntile * sizeof(int), 0));
  if (s->misC > 1) AQ_HIP(hipMemsetAsync(s->pflag, 0, (size_t)s->ntile * s->misC * sizeof(int), 0));
  size_t lds = (size_t)(8 * 256 + 11 * 256 + 5 * 256 + 8 * 4 * 16 + 2 * 256 * 17) * sizeof(double) + 16 * sizeof(int) +
               (size_t)16 * s->Mmax * sizeof(unsigned short);
#define AQ_MIS(NT_)                                                                                             \
  if (s->NT == NT_) {                                                                                          \
    AQ_HIP(hipFuncSetAttribute((const void *)aq_core_sweep_mis_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((aq_core_sweep_mis_kernel<NT_>), dim3(s->ntile * s->misC * nseg), dim3(512), lds, 0, t); \
  } else
  AQ_MIS(1) AQ_MIS(2) AQ_MIS(4) AQ_MIS(8) AQ_MIS(16) { return aq_fail(AQ_ERR_UNSUPPORTED, "no masked MFMA kernel instantiation for this n"); }
#undef AQ_MIS
  if (nseg > 1)
    hipLaunchKernelGGL(aq_k_combine_segment_sums6, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, s->sums, s->q_pad, nseg);
  if (s->misC > 1)
    hipLaunchKernelGGL(aq_k_sum_parts, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, s->rnpart, s->sums + (size_t)4 * s->q_pad, s->misC, s->q_pad);
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

static int aq_launch_core(aq_vb *s, int mode, double c) {
  AqCoreArgs a;
  a.XA = s->XA; a.XU = s->XU; a.G = s->G; a.Gx = s->Gx; a.R = s->R; a.gam = s->gam; a.mu = s->mu;
  a.theta = s->theta; a.zeta = s->zeta; a.sqrt_c = std::sqrt(c);
  a.c_is_one = std::fabs(c - 1.0) < 1.5e-8 ? 1 : 0;   // isTRUE(all.equal(c, 1)), R/update_vb.R:219
  a.Aarr = s->Aarr; a.Barr = s->Barr; a.coef = s->coef; a.inv2s = s->inv2s; a.cst = s->cst; a.sig2b = s->sig2b;
  a.sums = s->sums; a.rowGB = s->rowGB;
  a.c = c;
  a.p = s->p; a.q = s->q; a.p_pad = s->p_pad; a.q_pad = s->q_pad; a.n_pad = s->n_pad; a.nb = s->nb; a.ntile = s->ntile;
  a.dmode = s->dmode; a.mode = mode;
  hipEvent_t e0, e1;
  AQ_HIP(hipEventCreate(&e0));
  AQ_HIP(hipEventCreate(&e1));
  AQ_HIP(hipEventRecord(e0, 0));
  if (s->use_mis) {
    AQ_TRY(aq_launch_mis(s, mode, c));
  } else if (s->use_tw) {
    AQ_TRY(aq_launch_tw(s, mode, c));
  } else if (s->use_la) {
    const unsigned nwg = (unsigned)(s->ntile / s->TT);
    a.done = s->done; a.errflag = s->errflag; a.stagger = s->stagger;
    a.C = s->laC; a.xhelper = s->la_xhelper; a.Pbuf = s->Pbuf; a.pflag = s->pflag; a.rnpart = s->rnpart;
    a.xtouch = s->la_xtouch;
    a.hprio = s->la_hprio;
    a.mprio = s->la_mprio;
    a.mis = s->mis; a.GK = s->GK; a.tau = s->tau; a.log_tau = s->log_tau;
    a.sig2_inv_p = &s->sc->sig2_inv; a.log_sig2_inv_p = &s->sc->log_sig2_inv;
    if (s->laC > 1 && (!a.Pbuf || !a.rnpart || !a.errflag)) return aq_fail(AQ_ERR_DEVICE, "sample split without its exchange buffers");
    if (s->laC > 1 && mode == 0)   // the exchange slots start with tag 0 (aq_core_sweep_la.h, split_exchange)
      AQ_HIP(hipMemsetAsync(s->Pbuf, 0, (size_t)s->ntile * 2 * s->laC * 256 * sizeof(double), 0));
    a.dbg = nullptr;
    static long long *dbg_buf = nullptr;   // AQ_DIAG_DUMP=<file> with a -DAQ_DIAG_TIME build: per-role wait / total cycles of sweep 15
    const char *dump = getenv("AQ_DIAG_DUMP");
    const size_t dbg_n = (size_t)32 * nwg * 8 * 3 + 8 * 32 * 8;   // per-wave counters of up to 32 nwg workgroups + the timeline of workgroup 0
    if (dump && mode == 0) {
      static size_t dbg_cap = 0;
      if (dbg_n > dbg_cap) {
        if (dbg_buf) AQ_HIP(hipFree(dbg_buf));
        dbg_buf = nullptr;
        AQ_HIP(hipMalloc((void **)&dbg_buf, dbg_n * sizeof(long long)));
        dbg_cap = dbg_n;
      }
      AQ_HIP(hipMemsetAsync(dbg_buf, 0, dbg_n * sizeof(long long), 0));
      a.dbg = dbg_buf;
    }
    const bool chained = (mode == 0 && s->chain > 1);
    a.nseg = chained ? s->chain : 1;
    if (chained) AQ_HIP(hipMemsetAsync(s->done, 0, (size_t)s->ntile * sizeof(int), 0));
    // chained-segment launch: chain * nwg workgroups, workgroup s*nwg + k = SNP segment s of trait-tile group k
    const unsigned grid = chained ? (unsigned)((long long)s->chain * nwg) : nwg * (unsigned)s->laC;
    // One instance per handle for annealed and post-annealing sweeps alike: with the probit tables an annealed entry costs the
    // same three polynomials as any other (round 2 swapped to a (NT, NT, 3) geometry for the annealed sweeps).
    int lrc = s->la_mask ? aq_la_launch_mask(s->NT, s->NT2, s->NT3x, chained, grid, 0, a)
              : s->TT == 2 ? aq_la_launch_tt2(s->NT, s->NT2, s->NT3x, chained, grid, 0, a) : aq_la_launch_tt1(s->NT, s->NT2, s->NT3x, chained, grid, 0, a);
    if (lrc != 0) return aq_fail(AQ_ERR_UNSUPPORTED, "no look-ahead kernel instantiation for this n");
    if (chained) {
      if (s->la_mask) hipLaunchKernelGGL(aq_k_combine_segment_sums6, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, s->sums, s->q_pad, s->chain);
      else hipLaunchKernelGGL(aq_k_combine_segment_sums, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, s->sums, s->q_pad, s->chain);
    }
    if (s->laC > 1)
      hipLaunchKernelGGL(aq_k_sum_parts, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, s->rnpart, s->sums + (size_t)4 * s->q_pad, s->laC, s->q_pad);
    if (a.dbg && s->it == 15) {
      AQ_HIP(hipDeviceSynchronize());
      std::vector<long long> h((size_t)grid * 24 + 8 * 32 * 8);
      AQ_HIP(hipMemcpy(h.data(), a.dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      if (FILE *f = fopen(dump, "w")) {
        for (unsigned b = 0; b < grid; b++)
          for (int w = 0; w < 8; w++)
            fprintf(f, "%u %d %lld %lld %lld\n", b, w, h[((size_t)b * 8 + w) * 3], h[((size_t)b * 8 + w) * 3 + 1], h[((size_t)b * 8 + w) * 3 + 2]);
        fclose(f);
      }
      if (FILE *f = fopen((std::string(dump) + ".timeline").c_str(), "w")) {   // workgroup 0, phases 64 .. 95: wave, phase, 4 marks
        const long long *t = h.data() + (size_t)grid * 24;
        for (int w = 0; w < 8; w++)
          for (int i = 0; i < 32; i++)
            fprintf(f, "%d %d %lld %lld %lld %lld %lld %lld %lld\n", w, 64 + i, t[(w * 32 + i) * 8], t[(w * 32 + i) * 8 + 1], t[(w * 32 + i) * 8 + 2],
                    t[(w * 32 + i) * 8 + 3], t[(w * 32 + i) * 8 + 4], t[(w * 32 + i) * 8 + 5], t[(w * 32 + i) * 8 + 6]);
        fclose(f);
      }
    }
  } else {
    return aq_fail(AQ_ERR_UNSUPPORTED, "no core kernel selected for this problem");
  }
  AQ_HIP(hipEventRecord(e1, 0));
  AQ_HIP(hipGetLastError());
  if (mode == 0) {
    s->ev.push_back({e0, e1});
  } else {
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  return AQ_OK;
}

static AqQvec aq_qvec(aq_vb *s) {
  AqQvec v;
  v.eta_h = s->eta_h; v.kappa_h = s->kappa_h; v.n0 = s->n0; v.nobs = s->nobs;
  v.zeta = s->zeta; v.tau = s->tau; v.sig2b = s->sig2b; v.log_tau = s->log_tau; v.eta_vb = s->eta_vb;
  v.kappa_vb = s->kappa_vb; v.coef = s->coef; v.inv2s = s->inv2s; v.cst = s->cst; v.sums = s->sums;
  v.colApart = s->colApart; v.nchunk = s->fused ? 0 : s->nHchunk;   // fused: colSums(a) is already inside sums[3]
  v.q = s->q; v.q_pad = s->q_pad; v.n = s->n; v.nu_h = s->nu; v.rho_h = s->rho;
  v.na = s->has_missing ? 1 : 0;
  return v;
}
static AqPvec aq_pvec(aq_vb *s) {
  AqPvec v;
  v.theta = s->theta; v.sig2_theta = s->sig2_theta; v.L = s->L; v.lam2_inv = s->lam2_inv; v.Q = s->Q;
  v.rsZ = s->red; v.part = s->ppart; v.p = s->p; v.p_pad = s->p_pad; v.shr = s->shr; v.m0 = s->m0;
  v.A2_inv = s->A2_inv; v.df = (double)s->df;
  return v;
}

extern "C" int aq_vb_create(const aq_vb_problem *pr, aq_vb_handle *out) {
  if (!pr || !out) return aq_fail(AQ_ERR_ARG, "aq_vb_create: NULL argument");
  *out = nullptr;
  if (pr->n < 2 || pr->p < 1 || pr->q < 1) return aq_fail(AQ_ERR_ARG, "aq_vb_create: n >= 2, p >= 1, q >= 1 required");
  if (pr->q_total < pr->q) return aq_fail(AQ_ERR_ARG, "aq_vb_create: q_total < q");
  if (!pr->X || !pr->Y || !pr->eta || !pr->kappa || !pr->n0 || (!pr->init_generate && (!pr->gam_vb || !pr->mu_beta_vb)) || !pr->sig2_beta_vb ||
      !pr->sig2_theta_vb || !pr->tau_vb || !pr->theta_vb || !pr->zeta_vb)
    return aq_fail(AQ_ERR_ARG, "aq_vb_create: NULL data pointer");
  if (!(pr->tol > 0)) return aq_fail(AQ_ERR_ARG, "tol must be positive");
  if (pr->maxit < 1) return aq_fail(AQ_ERR_ARG, "maxit must be natural.");
  if (pr->has_anneal) {   // check_annealing_, R/prepare_atlasqtl.R:100-124
    int type = (int)std::llround(pr->anneal[0]);
    if (type < 1 || type > 3)
      return aq_fail(AQ_ERR_ARG, "The annealing spacing scheme must be set to 1 for geometric 2 for harmonic or 3 for linear spacing.");
    if (pr->anneal[1] < 1.5) return aq_fail(AQ_ERR_ARG, "Initial annealing temperature very small.");
    if (pr->anneal[2] > 1000 || pr->anneal[2] < 2) return aq_fail(AQ_ERR_ARG, "Temperature grid size out of range.");
  }
  if (pr->world_size < 1) return aq_fail(AQ_ERR_ARG, "world_size must be >= 1");
  if (pr->scheme != 0 && pr->scheme != 1) return aq_fail(AQ_ERR_ARG, "scheme must be 0 (global-local horseshoe) or 1 (global-only)");
  const int df = pr->df == 0 ? 1 : pr->df;
  if (pr->scheme == 0 && df != 1 && df != 3 && df != 5 && df != 7)
    return aq_fail(AQ_ERR_UNSUPPORTED, "df must be 1, 3, 5 or 7 (the reference calls compute_integral_hs_ unstable from df = 9 on, R/utils.R:510)");
  AQ_TRY(aq_need_device(pr->device));

  // X must be complete; Y may hold NaN.  With xy_on_device both are device pointers: X is trusted to be the standardised
  // NaN-free matrix of aq_prepare_data, Y (n x q, small) is copied back once for the missingness bookkeeping below.
  size_t np = (size_t)pr->n * pr->p, nq = (size_t)pr->n * pr->q;
  std::vector<double> Yhost;
  const double *Yh = pr->Y;
  const bool x_dev = (pr->xy_on_device & 1) != 0, y_dev = (pr->xy_on_device & 2) != 0;
  if (y_dev) {
    Yhost.resize(nq);
    AQ_HIP(hipMemcpy(Yhost.data(), pr->Y, nq * sizeof(double), hipMemcpyDeviceToHost));
    Yh = Yhost.data();
  }
  if (!x_dev) {
    for (size_t i = 0; i < np; i++)
      if (!(pr->X[i] == pr->X[i])) return aq_fail(AQ_ERR_ARG, "X must be a non-empty a numeric matrix, finite without missing value.");
  }
  bool has_missing = false;
  for (size_t i = 0; i < nq && !has_missing; i++) has_missing = !(Yh[i] == Yh[i]);
  if (pr->n > 10240)
    return aq_fail(AQ_ERR_UNSUPPORTED, "n > 10240: a trait's residual no longer fits the registers of 4 waves (not implemented yet)");

  aq_vb *s = new aq_vb();
  s->device = pr->device;
  s->n = pr->n; s->p = pr->p; s->q = pr->q; s->q_total = pr->q_total; s->world = pr->world_size;
  s->trait_offset = pr->trait_offset;
  s->p_pad = (pr->p + 15) / 16 * 16;
  s->q_pad = (pr->q + 15) / 16 * 16;
  s->nb = s->p_pad / 16;
  s->ntile = s->q_pad / 16;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, pr->device) == hipSuccess && prop.multiProcessorCount > 0) s->ncu = prop.multiProcessorCount;
    if (const char *e = aq_env(s, "AQ_NCU")) s->ncu = atoi(e) > 0 ? atoi(e) : s->ncu;
  }
  {
    // default: look-ahead kernel (complete Y, n <= 1056); AQ_KERNEL=2 forces the generic wave-per-trait kernel
    const char *ek = aq_env(s, "AQ_KERNEL");
    // missing values: masked blocked MFMA kernel while n fits 8 waves x 16 residual tiles and no trait misses more
    // than AQ_MIS_MMAX samples; otherwise (and with AQ_KERNEL=2) the generic kernel
    int max_missing = 0;
    if (has_missing) {
      for (int k = 0; k < pr->q; k++) {
        int m = 0;
        for (int i = 0; i < pr->n; i++) m += !(Yh[(size_t)i + (size_t)pr->n * k] == Yh[(size_t)i + (size_t)pr->n * k]);
        if (m > max_missing) max_missing = m;
        s->max_missing = max_missing;
      }
    }
    // the masked MFMA kernel also serves complete Y beyond the look-ahead kernel's n (all-ones mask, empty lists)
    // Y with missing values: the look-ahead kernel's MASK instances when the per-trait Gram blocks (50 KB per trait tile and
    // SNP block, computed once) fit next to the rest of the state in HBM; else the masked two-barrier kernel (AQ_KERNEL=3
    // forces that one), which recomputes them every sweep.  Complete Y beyond n = 1056: look-ahead kernel with a sample split.
    const bool n_la_ok = pr->n <= 8 * 16 * 105;
    bool la_mask_ok = has_missing && n_la_ok && max_missing <= AQ_MIS_MMAX && !(ek && atoi(ek) >= 2);
    if (la_mask_ok) {
      size_t free_b = 0, tot_b = 0;
      const size_t ntile_ = (size_t)(pr->q + 15) / 16, nb_ = (size_t)(pr->p + 15) / 16;
      const size_t gk_b = ntile_ * nb_ * AQ_GK_STRIDE * sizeof(double);
      const size_t rest_b = 2 * ntile_ * nb_ * 256 * sizeof(double) + 3 * (size_t)(pr->n + 64) * nb_ * 16 * sizeof(double) * 2 +
                            3 * ntile_ * (size_t)(pr->n + 64) * 16 * sizeof(double);
      // decided on the device's TOTAL memory (minus a tenth), not on what happens to be free: the same problem gets the same
      // kernel on every rank and in every run, so a checkpoint of one is accepted by the other.  Should the allocation then fail
      // because other processes hold memory, aq_vb_create reports the out-of-memory error (AQ_GK_MAX_GB lowers the limit).
      if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess || (double)(gk_b + rest_b) * 1.05 > 0.9 * (double)tot_b) la_mask_ok = false;
      if (const char *e = aq_env(s, "AQ_GK_MAX_GB")) if ((double)gk_b > atof(e) * 1e9) la_mask_ok = false;   // test hook: force the fallback
    }
    const bool la_split_ok = !has_missing && pr->n > 1056 && n_la_ok && !(ek && atoi(ek) >= 2);   // complete Y, large n
    if (!la_mask_ok && (has_missing || (pr->n > 1056 && !la_split_ok)) && pr->n <= 16384 && max_missing <= AQ_MIS_MMAX && !(ek && atoi(ek) == 2)) {
      s->use_mis = true;
      s->NW = 8;
      // n_pad = 128 NT C: C workgroups per trait tile, NT in {1,2,4,8,16} residual tiles per wave.  Model of a sweep:
      // whole rounds of workgroups (one per CU) x time per SNP block (4.5 + 0.94 NT us, + 6 us for the exchange)
      {
        const int ntile_ = (pr->q + 15) / 16;
        double best = 1e300;
        for (int C = 1; C <= 8; C++)
          for (int NT = 1; NT <= 16; NT *= 2) {
            if (128 * NT * C < pr->n) continue;
            double rounds = (double)(((long long)ntile_ * C + s->ncu - 1) / s->ncu);
            double cost = rounds * (4.5 + 0.94 * NT + (C > 1 ? 6.0 : 0.0));
            if (cost < best - 1e-9) { best = cost; s->misC = C; s->NT = NT; }
          }
      }
      if (const char *e = aq_env(s, "AQ_MIS_C")) {   // test hook: force the sample split at small n
        int C = atoi(e);
        if (C >= 1 && C <= 8) {
          s->misC = C;
          s->NT = 16;
          for (int NT = 16; NT >= 1; NT /= 2) if (128 * NT * C >= pr->n) s->NT = NT;
        }
      }
      s->Mmax = (max_missing + 15) / 16 * 16;
      if (s->Mmax < 16) s->Mmax = 16;
    } else if ((has_missing && !la_mask_ok) || (ek && atoi(ek) == 2) || (pr->n > 1056 && !la_split_ok && !la_mask_ok)) {
      // generic kernel geometry: n_pad = 64 * NE * WPT samples, WPT waves (and workgroups) per trait (tile)
      s->use_tw = true;
      s->WPT = pr->n <= 2048 ? 1 : pr->n <= 5120 ? 2 : 4;
      if (const char *e = aq_env(s, "AQ_TW_WPT")) { int v = atoi(e); if ((v == 2 || v == 4) && v > s->WPT) s->WPT = v; }   // test hook
      const int per_lane = (pr->n + 64 * s->WPT - 1) / (64 * s->WPT);
      s->NE = per_lane <= 4 ? 4 : per_lane <= 8 ? 8 : per_lane <= 16 ? 16 : per_lane <= 32 ? 32 : 40;
    }
    if (!s->use_tw && !s->use_mis) {
      // 6 matrix waves (NT tiles on waves 0-2, NT2 = NT or NT - 1 on waves 4-6: NT + NT2 per SIMD) + the recurrence wave,
      // which owns aq_la_nt3(NT, TT) tiles of its own when two trait tiles share a workgroup
      s->use_la = true;
      s->la_mask = la_mask_ok;
      const int ntiles = (pr->n + 15) / 16;
      s->NW = 6;
      // two trait tiles per workgroup once that still gives every CU a workgroup (C3: 625 tiles -> 313 workgroups of 32
      // traits): X operands shared by two MFMAs, one chain evaluation per 32 traits, half the per-phase overhead.
      // q is then padded to a multiple of 32 (the extra tile is all padding: zero residual, masked sums).
      // (Crossover measured at n = 1000, 256 CUs: 448 tiles -- one round of two-tile workgroups, 27.4 ms, against 1.75 rounds of
      // one-tile workgroups; q = 8000: 27.5 vs 31.1 ms, q = 6144: 27.5 vs 23.4.)
      s->TT = (4LL * s->ntile >= 7LL * s->ncu) ? 2 : 1;
      if (const char *e = aq_env(s, "AQ_TT")) s->TT = atoi(e) == 2 ? 2 : 1;
      if (s->la_mask) s->TT = 1;   // 16 per-trait Gram blocks per trait tile in LDS: one tile per workgroup
      if (s->TT == 2) {
        s->q_pad = (pr->q + 31) / 32 * 32;
        s->ntile = s->q_pad / 16;
        // (waves 4-6 then enter a phase when their SIMD partner is a third of the way through it -- stagger, set below -- so
        // that one wave's hand-off gap is covered by the other's MFMAs)
      }
      {
        // smallest geometry that holds ntiles: NT in 1..11, NT2 in {NT, NT - 1}, plus the recurrence wave's aq_la_nt3 tiles;
        // among equals the one with more tiles on the recurrence wave (AQ_NT3=0/3/6 pins its tile count for experiments)
        const char *e3 = aq_env(s, "AQ_NT3");
        auto fit = [&](int tiles_needed, int nt_max, int *NTo, int *NT2o, int *N3xo) {
          int best_tiles = 1 << 30, best_nt3 = -1;
          for (int NT = 1; NT <= nt_max; NT++)
            for (int NT2 = NT; NT2 >= (NT > 1 ? NT - 1 : NT); NT2--) {
              // x9: the instance NT / NT / 9 -- nine residual tiles on the recurrence wave, 18 instead of 19 per matrix SIMD at
              // n = 1000.  While the recurrence wave's MFMAs hold SIMD 3's datapath the helper wave's fp64 work crawls (38.4 ms at C3
              // against 34.3 for 10 / 9 / 6), so it comes with the helper wave one priority level up (set below): 33.9 ms
              // (profiles/r03_nt9_stagger.txt).  AQ_NT3 pins another count; the diagnostic build, whose NT / NT / 9 instances do not
              // pass the ISA proof, takes it only on request.
#ifdef AQ_DIAG_TIME
              const bool x9_ok = s->TT == 2 && NT >= 8 && NT2 == NT && e3 && atoi(e3) == 9;
#else
              // (by default from NT = 9 on: 8 / 8 / 9 -- n around 900 -- loses to 9 / 8 / 6, 8.68 against 8.42 us per phase; measured per
              // geometry in profiles/r03_nt9_stagger.txt: a phase takes max(matrix SIMDs, SIMD 3) with SIMD 3 at 7.6 - 7.7 us for three or
              // six tiles and 8.7 for nine, the matrix SIMDs at 7.6 / 7.7 / 8.4 / 8.4 / 8.9 / 9.3 / 9.7 / 10.3 us for (8,7) ... (11,11))
              const bool x9_ok = s->TT == 2 && NT2 == NT && (e3 ? (NT >= 8 && atoi(e3) == 9) : NT >= 9);
#endif
              for (int x9 = 0; x9 <= (x9_ok ? 1 : 0); x9++) {
                int nt3 = x9 ? 9 : aq_la_nt3(NT, NT2, s->TT);
                // one tile per workgroup, unsplit (the trait shards of N = 2, 4: MFMA-bound on three SIMDs while SIMD 3 only runs
                // the chain): three residual tiles on the recurrence wave by default -- q = 5000: 19.55 -> 18.78 ms, q = 2500:
                // 15.48 -> 14.83; six or nine make its chain + tiles the bound (30 and 33 ms).  AQ_NT3 = 0 / 3 / 6 / 9 pins the count.
                int x1 = -1;
                bool second = false;
                if (s->TT == 1 && !s->la_mask && NT >= 8 && nt_max <= 11) {
                  const int w = e3 ? atoi(e3) : 3;
                  if ((w == 3 || w == 9) && NT2 == NT) x1 = w;
                  else if (w == 6 && NT2 == NT - 1) x1 = w;
                  else if (e3 && w != 0) continue;
                  if (x1 > 0) nt3 = x1;
                  second = !e3 && x1 > 0;        // by default the plain geometry competes as well (it wins when it needs fewer tiles)
                }
                if (second) {
                  const int tiles0 = 3 * (NT + NT2);
                  if (tiles0 >= tiles_needed && tiles0 < best_tiles) { best_tiles = tiles0; best_nt3 = 0; *NTo = NT; *NT2o = NT2; *N3xo = -1; }
                }
                const int tiles = 3 * (NT + NT2) + nt3;
                if (tiles < tiles_needed || (e3 && atoi(e3) != nt3 && s->TT == 2 && NT >= 8)) continue;
                if (tiles < best_tiles || (tiles == best_tiles && nt3 > best_nt3)) { best_tiles = tiles; best_nt3 = nt3; *NTo = NT; *NT2o = NT2; *N3xo = x9 ? 9 : x1; }
              }
            }
          return best_tiles;
        };
        if (pr->n <= 1056 && !aq_env(s, "AQ_LA_C")) {
          s->laC = 1;
          const int tiles = fit(ntiles, 11, &s->NT, &s->NT2, &s->NT3x);    // n <= 1056 always fits (11, 11) ...
          if (tiles >= (1 << 30)) { delete s; return aq_fail(AQ_ERR_ARG, "AQ_NT3 excludes every look-ahead geometry for this n"); }   // ... unless the test hook forbids it
          s->n_pad = 16 * tiles;
          // Few trait groups (a trait shard of a multi-GPU run, a small q): the CUs left idle share the samples.  Per SNP block an
          // unsplit workgroup needs its MFMA stream (0.213 us per residual tile of one SIMD + 1.0 of hand-offs; n = 1000: 5.46 us
          // measured) or, for small n, the chain (3.3 us); a part of a split group needs its shorter stream or chain + exchange
          // (4.5 us with two parts, measured at n = 1000: 4.53; + 0.2 per further part).  All parts must run at once.
          // (Not with missing values: there the chain is longer and the helper wave loaded -- q = 1250 with 5 % NA: 19.3 ms unsplit,
          // 20.1 split.  AQ_LA_NOSPLIT=1 keeps one workgroup per group: experiments.)
          if (s->TT == 1 && !s->la_mask && !aq_env(s, "AQ_LA_NOSPLIT")) {
            double best = std::max(0.213 * (s->NT + s->NT2) + 1.0, 3.3) * 0.95;   // a split must win by 5 %
            for (int C = 2; C <= 8 && (long long)s->ntile * C <= s->ncu; C++) {
              int NT = 0, NT2 = 0, N3x = -1;
              const int tiles_c = fit((ntiles + C - 1) / C, 18, &NT, &NT2, &N3x);
              if (tiles_c >= (1 << 30)) continue;
              const double cost = std::max(0.213 * (NT + NT2) + 1.0, 4.5 + 0.2 * (C - 2));
              if (cost < best - 1e-9) { best = cost; s->laC = C; s->NT = NT; s->NT2 = NT2; s->NT3x = N3x; s->n_pad = 16 * tiles_c * C; }
            }
          }
        } else {
          // n beyond one workgroup's registers: C workgroups share a trait group (sample split, one tile per workgroup).  Cost of
          // a sweep ~ rounds of workgroups x time per SNP block: the MFMA stream of one SIMD (0.213 us per residual tile) or the
          // exchange + chain (4.5 us with two parts, + 0.2 per further part), whichever is longer.  (AQ_LA_C forces the split at small n: test hook.)
          s->TT = 1; s->q_pad = (pr->q + 15) / 16 * 16; s->ntile = s->q_pad / 16; s->stagger = 0;
          double best = 1e300;
          const char *ec = aq_env(s, "AQ_LA_C");
          for (int C = 2; C <= 8; C++) {
            if (ec && atoi(ec) != C) continue;
            int NT = 0, NT2 = 0, N3x = -1;
            const int tiles = fit((ntiles + C - 1) / C, 18, &NT, &NT2, &N3x);
            if (tiles >= (1 << 30)) continue;
            const double rounds = (double)(((long long)s->ntile * C + s->ncu - 1) / s->ncu);
            const double cost = rounds * std::max(0.213 * (NT + NT2) + 1.0, 4.5 + 0.2 * (C - 2));
            if (cost < best - 1e-9) { best = cost; s->laC = C; s->NT = NT; s->NT2 = NT2; s->NT3x = N3x; s->n_pad = 16 * tiles * C; }
          }
          if (best >= 1e300) { delete s; return aq_fail(AQ_ERR_UNSUPPORTED, "no look-ahead geometry for this n"); }
          // Who exchanges the partial S': the recurrence wave at the start of its chain.  The helper wave can do it a block ahead
          // (AQ_LA_XHELPER=1); that paid at n = 5000 while an exchange was three trips through the shared cache (236 vs 225 ms),
          // with self-validating words it no longer does (222.5 vs 222.0; n = 1500: 58.3 vs 64.5).
          s->la_xhelper = 0;
        }
        if (const char *e = aq_env(s, "AQ_LA_XHELPER")) s->la_xhelper = atoi(e) != 0;   // test hook
        if (const char *e = aq_env(s, "AQ_XTOUCH")) s->la_xtouch = atoi(e) != 0;
        // helper wave one priority level above the recurrence wave it shares SIMD 3 with -- only where its iteration is on the
        // critical cycle: the unsplit MASK instances (single cross-block buffer; C3 + 5 % NA 49.9 -> 48.6 ms on one box, 48.4 -> 47.4
        // on another; the split C5 shard 241.1 -> 242.0, complete Y the same within noise: profiles/r03_hprio_na.txt)
        if (s->la_mask && s->laC <= 1) s->la_hprio = 1;
        // six or nine tiles on the recurrence wave of a two-tile workgroup: see fit() (n = 800, geometry 8 / 7 / 6: 33.5 -> 29.6 ms)
        if (s->TT == 2 && s->NT >= 8 && (s->NT3x == 9 || (s->NT3x < 0 && s->NT2 == s->NT - 1))) s->la_hprio = 1;
        if (const char *e = aq_env(s, "AQ_MPRIO")) s->la_mprio = atoi(e) != 0;
        if (const char *e = aq_env(s, "AQ_HPRIO")) s->la_hprio = atoi(e) >= 0 && atoi(e) <= 3 ? atoi(e) : 0;
        s->la_nt3_pinned = aq_env(s, "AQ_NT3") != nullptr;
        if (s->TT == 2) s->stagger = (s->NT + 2) / 3;
      }
      if (const char *e = aq_env(s, "AQ_STAGGER")) s->stagger = atoi(e) >= 0 ? atoi(e) : 0;
      // more workgroups than CUs: chained SNP segments even out the last round (3 rounds -> ~2.5 for 625 workgroups)
      const int nwg = s->ntile / s->TT;
      if (nwg > s->ncu && s->laC == 1) {
        // cost of a sweep in phases: rounds of workgroups x (SNP blocks of a segment + 2 phases of pipeline fill + ~1.7 of start-up:
        // fitted to C3 at S = 4, 8, 13, 16, 26, profiles/r03_chain_segments.txt).  C3: 313 groups -> S = 13 (16 rounds of 241 blocks:
        // 34.43 ms against 34.61 with S = 4); C3 with NA: 625 groups -> S = 9.
        double best = 1e30;
        for (int S = 2; S <= 32 && S <= s->nb; S++) {
          const long long wg = (long long)nwg * S;
          const double cost = (double)((wg + s->ncu - 1) / s->ncu) * ((s->nb + S - 1) / S + 3.7);
          if (cost < best - 1e-9) { best = cost; s->chain = S; }
        }
        if (best >= (double)((nwg + s->ncu - 1) / s->ncu) * (s->nb + 3.7)) s->chain = 0;   // no gain over whole tiles
      }
      if (const char *e = aq_env(s, "AQ_CHAIN")) s->chain = atoi(e) > 1 ? atoi(e) : 0;
      if (s->chain > s->nb) s->chain = s->nb;
      if (s->chain > 32) s->chain = 32;
      if (s->laC > 1) s->chain = 0;   // the parts of a group must be co-resident: no chained segments
    }
  }
  if (s->use_mis && s->misC == 1) {   // same chained-segment choice as for the look-ahead kernel
    if (s->ntile > s->ncu) {
      double best = 1e30;
      for (int S = 2; S <= 16; S++) {
        long long wg = (long long)s->ntile * S;
        double cost = (double)((wg + s->ncu - 1) / s->ncu) / S * (1.0 + 0.002 * S);
        if (cost < best - 1e-12) { best = cost; s->chain = S; }
      }
      if (best >= (double)((s->ntile + s->ncu - 1) / s->ncu)) s->chain = 0;
    }
    if (const char *e = aq_env(s, "AQ_CHAIN")) s->chain = atoi(e) > 1 ? atoi(e) : 0;
    if (s->chain > s->nb) s->chain = s->nb;
    if (s->chain > 32) s->chain = 32;
  }
  if (s->use_mis) { s->n_pad = 128 * s->NT * s->misC; s->NR = s->n_pad + 8; }
  if (s->use_tw) {
    s->n_pad = 64 * s->NE * s->WPT;
    // SNP columns staged in LDS at a time: ns * n_pad doubles next to 18 KB of block scalars, within 160 KB
    s->tw_ns = 4;
    while (s->tw_ns > 1 && (size_t)(s->tw_ns * s->n_pad + 8 * 256 + 32) * sizeof(double) > 150 * 1024) s->tw_ns /= 2;
  }
  int rc = aq_probe_dmode(&s->dmode);
  if (rc != AQ_OK) { delete s; return rc; }
  if ((s->use_la || s->use_mis) && s->dmode != 0) {   // the MFMA sweep kernels are written for gfx950's accumulator map (row = 4 reg + lane / 16)
    delete s;
    return aq_fail(AQ_ERR_UNSUPPORTED, "this device reports an f64 MFMA accumulator layout the sweep kernels are not written for");
  }

  s->A2_inv = pr->A2_inv; s->m0 = pr->m0; s->nu = pr->nu; s->rho = pr->rho; s->t02 = pr->t02;
  s->t02_inv = 1.0 / pr->t02;
  s->shr = (double)pr->q_total;   // shr_fac_inv <- q, R/atlasqtl.R:218
  s->has_anneal = pr->has_anneal != 0;
  s->scheme = pr->scheme; s->df = pr->scheme == 1 ? 1 : (pr->df == 0 ? 1 : pr->df);
  std::memcpy(s->anneal, pr->anneal, sizeof(s->anneal));
  s->tol = pr->tol; s->maxit = pr->maxit; s->thinned = pr->thinned_elbo_eval != 0; s->debug = pr->debug != 0;
  s->has_missing = has_missing || s->use_mis || s->la_mask;   // the masked kernel produces the NA forms of the column sums (identical for complete Y)

  auto fail = [&](int code) { aq_free_all(s); return code; };
#define AQ_TRYF(x) do { int rc2_ = (x); if (rc2_ != AQ_OK) return fail(rc2_); } while (0)
#define AQ_HIPF(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { aq_fail(AQ_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); return fail(AQ_ERR_DEVICE); } } while (0)

  const int NTT = s->n_pad / 16;
  size_t xelems = s->use_tw ? 1 : (size_t)s->nb * NTT * 128;
  AQ_TRYF(aq_dalloc(&s->XA, xelems));
  AQ_TRYF(aq_dalloc(&s->XU, xelems));
  AQ_TRYF(aq_dalloc(&s->G, (size_t)s->nb * 256));
  AQ_TRYF(aq_dalloc(&s->Gx, (size_t)s->nb * 256));
  if (s->use_tw) {
    AQ_TRYF(aq_dalloc(&s->mis, (size_t)s->ntile * s->n_pad * 16));
    AQ_TRYF(aq_dalloc(&s->XN, (size_t)s->ntile * s->p_pad * 16));
  }
  if (s->la_mask) {
    s->Mmax = (s->max_missing + 15) / 16 * 16;
    if (s->Mmax < 16) s->Mmax = 16;
    s->NR = s->n_pad + 8;
  }
  if (s->use_mis || s->la_mask) {
    AQ_TRYF(aq_dalloc(&s->mis, (size_t)s->ntile * s->n_pad * 16));
    AQ_TRYF(aq_dalloc(&s->XR, (size_t)s->nb * s->NR * 16));
    AQ_TRYF(aq_dalloc(&s->midx, (size_t)s->ntile * 16 * s->Mmax));
    AQ_TRYF(aq_dalloc(&s->mcnt4, (size_t)s->ntile * 16));
    if (s->use_mis) {
      AQ_TRYF(aq_dalloc(&s->Pbuf, (size_t)s->ntile * 2 * s->misC * 256));
      AQ_TRYF(aq_dalloc(&s->pflag, (size_t)s->ntile * s->misC));
      AQ_TRYF(aq_dalloc(&s->rnpart, (size_t)s->misC * s->q_pad));
    }
    // lists of missing samples per trait, padded to groups of 16 with the all-zero row n_pad of XR
    std::vector<int> idx((size_t)s->ntile * 16 * s->Mmax, s->n_pad), cnt((size_t)s->ntile * 16, 0);
    for (int k = 0; k < s->q; k++) {
      int m = 0;
      int *dst = idx.data() + (size_t)k * s->Mmax;      // trait k = tile (k / 16), slot (k % 16): contiguous
      for (int i = 0; i < s->n; i++)
        if (!(Yh[(size_t)i + (size_t)s->n * k] == Yh[(size_t)i + (size_t)s->n * k])) dst[m++] = i;
      cnt[k] = (m + 15) / 16 * 4;
    }
    AQ_HIPF(hipMemcpy(s->midx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    AQ_HIPF(hipMemcpy(s->mcnt4, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (s->use_la && s->laC > 1) {
    AQ_TRYF(aq_dalloc(&s->Pbuf, (size_t)s->ntile * 2 * s->laC * 256));
    AQ_TRYF(aq_dalloc(&s->pflag, (size_t)s->ntile * s->laC));
    AQ_TRYF(aq_dalloc(&s->rnpart, (size_t)s->laC * s->q_pad));
  }
  AQ_TRYF(aq_dalloc(&s->R, (size_t)s->ntile * s->n_pad * 16));
  AQ_TRYF(aq_dalloc(&s->gam, (size_t)s->ntile * s->p_pad * 16));
  AQ_TRYF(aq_dalloc(&s->mu, (size_t)s->ntile * s->p_pad * 16));
  AQ_TRYF(aq_dalloc(&s->theta, (size_t)s->p_pad));
  AQ_TRYF(aq_dalloc(&s->sig2_theta, (size_t)s->p_pad));
  AQ_TRYF(aq_dalloc(&s->L, (size_t)s->p_pad));
  AQ_TRYF(aq_dalloc(&s->lam2_inv, (size_t)s->p_pad));
  AQ_TRYF(aq_dalloc(&s->Q, (size_t)s->p_pad));
  s->pblk = (s->p + 255) / 256;
  AQ_TRYF(aq_dalloc(&s->ppart, (size_t)3 * s->pblk));
  double **qv[] = {&s->eta_h, &s->kappa_h, &s->n0, &s->nobs, &s->zeta, &s->tau, &s->sig2b, &s->log_tau, &s->eta_vb,
                   &s->kappa_vb, &s->coef, &s->inv2s, &s->cst};
  for (double **qp : qv) AQ_TRYF(aq_dalloc(qp, (size_t)s->q_pad));
  AQ_TRYF(aq_dalloc(&s->sums, (size_t)6 * s->q_pad * (s->chain + 2)));   // one slot of 5 (look-ahead) or 6 (NA forms) rows per chained segment
  AQ_TRYF(aq_dalloc(&s->done, (size_t)s->ntile));
  AQ_TRYF(aq_dalloc(&s->errflag, (size_t)1));
  s->fused = s->use_la;   // the look-ahead kernel computes A, b and the sums of a itself: no pre-pass arrays
  if (!s->fused) AQ_TRYF(aq_dalloc(&s->rowA, (size_t)s->ntile * s->p_pad));
  AQ_TRYF(aq_dalloc(&s->rowGB, (size_t)s->ntile * s->WPT * s->p_pad));
  if (!s->fused) {   // the other kernels read A and b from the pre-pass arrays
    AQ_TRYF(aq_dalloc(&s->Aarr, (size_t)s->ntile * s->p_pad * 16));
    AQ_TRYF(aq_dalloc(&s->Barr, (size_t)s->ntile * s->p_pad * 16));
  }
  if (pr->ext_reduce_main) { s->red = pr->ext_reduce_main; s->own_red = false; }
  else { AQ_TRYF(aq_dalloc(&s->red, (size_t)aq_vb_reduce_len(s->p))); s->own_red = true; }
  if (pr->ext_reduce_elbo) { s->ered = pr->ext_reduce_elbo; s->own_ered = false; }
  else { AQ_TRYF(aq_dalloc(&s->ered, (size_t)8)); s->own_ered = true; }
  s->rows_per_chunk = 2048;
  s->nHchunk = (s->p_pad + s->rows_per_chunk - 1) / s->rows_per_chunk;
  AQ_TRYF(aq_dalloc(&s->Hpart, (size_t)s->ntile * s->nHchunk));
  AQ_TRYF(aq_dalloc(&s->colApart, (size_t)s->nHchunk * s->q_pad));
  AQ_TRYF(aq_dalloc(&s->sc, (size_t)1));

  // ---- uploads + layout conversion (staging buffers freed afterwards) ----
  {
    double *Xd = nullptr;
    const bool own_x = !x_dev || s->use_tw;   // the generic kernel keeps X: it needs a copy of its own
    if (own_x) {
      AQ_HIPF(hipMalloc((void **)&Xd, np * sizeof(double)));
      AQ_HIPF(hipMemcpy(Xd, pr->X, np * sizeof(double), x_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
      Xd = const_cast<double *>(pr->X);
    }
    if (s->use_tw) {
      s->Xcm = Xd;   // the generic kernel reads X column-major as given
    } else {
      hipLaunchKernelGGL(aq_k_build_x_layouts, dim3((unsigned)((xelems + 255) / 256)), dim3(256), 0, 0, Xd, s->XA, s->XU,
                         s->n, s->p, s->nb, NTT, s->dmode);
      hipLaunchKernelGGL(aq_k_gram_blocks, dim3(s->nb), dim3(256), 0, 0, Xd, s->G, s->Gx, s->n, s->p);
      if (s->use_mis || s->la_mask) {
        size_t tot = (size_t)s->nb * s->NR * 16;
        hipLaunchKernelGGL(aq_k_build_xr, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, Xd, s->XR, s->n, s->p, s->nb, s->NR);
      }
      AQ_HIPF(hipDeviceSynchronize());
      if (own_x) AQ_HIPF(hipFree(Xd));
      if (s->la_mask) {
        // the traits' own Gram blocks, once per handle (aq_core_sweep_mis.h::aq_k_gk_blocks); the row panels are not needed after
        AQ_TRYF(aq_dalloc(&s->GK, (size_t)s->ntile * s->nb * AQ_GK_STRIDE));
        const int bchunk = 32;
        const size_t lds = (size_t)16 * s->Mmax * sizeof(unsigned short);
        hipLaunchKernelGGL(aq_k_gk_blocks, dim3((s->nb + bchunk - 1) / bchunk, s->ntile), dim3(512), lds, 0, s->XR, s->G, s->Gx, s->midx, s->mcnt4,
                           s->GK, s->nb, s->NR, s->Mmax, bchunk);
        AQ_HIPF(hipGetLastError());
        AQ_HIPF(hipDeviceSynchronize());
        AQ_HIPF(hipFree(s->XR));
        s->XR = nullptr;
      }
    }
  }
  {
    size_t big = (pr->init_on_device || pr->init_generate) ? nq : std::max((size_t)pr->p * pr->q, nq);
    double *stage = nullptr;
    AQ_HIPF(hipMalloc((void **)&stage, big * sizeof(double)));
    AQ_HIPF(hipMemcpy(stage, pr->Y, nq * sizeof(double), y_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    hipLaunchKernelGGL(aq_k_tile_from_colmajor, dim3((s->n_pad + 63) / 64, s->ntile), dim3(256), 0, 0, stage, s->R,
                       s->n, s->q, s->n_pad, 1);
    AQ_HIPF(hipDeviceSynchronize());
    if (s->use_tw || s->use_mis || s->la_mask) {   // mis_pat <- ifelse(is.na(Y), 0, 1), R/atlasqtl_global_local_core.R:21
      std::vector<double> mk(nq);
      for (size_t i = 0; i < nq; i++) mk[i] = (Yh[i] == Yh[i]) ? 1.0 : 0.0;
      AQ_HIPF(hipMemcpy(stage, mk.data(), nq * sizeof(double), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(aq_k_tile_from_colmajor, dim3((s->n_pad + 63) / 64, s->ntile), dim3(256), 0, 0, stage, s->mis,
                         s->n, s->q, s->n_pad, 0);
      AQ_HIPF(hipDeviceSynchronize());
    }
    size_t pq = (size_t)pr->p * pr->q;
    const double *gsrc = pr->gam_vb, *msrc = pr->mu_beta_vb;
    if (pr->init_generate) {
      if (!(pr->init_gam_sd > 0.0)) { AQ_HIPF(hipFree(stage)); aq_fail(AQ_ERR_ARG, "init_gam_sd must be positive"); return fail(AQ_ERR_ARG); }
      size_t tot = (size_t)s->ntile * s->p_pad * 16;
      hipLaunchKernelGGL(aq_k_init_generate, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, s->gam, s->mu, s->p, s->q, s->p_pad,
                         s->ntile, (unsigned long long)pr->init_seed, (int)pr->trait_offset, pr->init_gam_mean, pr->init_gam_sd);
      AQ_HIPF(hipDeviceSynchronize());
    } else {
    if (!pr->init_on_device) {
      AQ_HIPF(hipMemcpy(stage, pr->gam_vb, pq * sizeof(double), hipMemcpyHostToDevice));
      gsrc = stage;
    }
    hipLaunchKernelGGL(aq_k_tile_from_colmajor, dim3((s->p_pad + 63) / 64, s->ntile), dim3(256), 0, 0, gsrc, s->gam,
                       s->p, s->q, s->p_pad, 0);
    AQ_HIPF(hipDeviceSynchronize());
    if (!pr->init_on_device) {
      AQ_HIPF(hipMemcpy(stage, pr->mu_beta_vb, pq * sizeof(double), hipMemcpyHostToDevice));
      msrc = stage;
    }
    hipLaunchKernelGGL(aq_k_tile_from_colmajor, dim3((s->p_pad + 63) / 64, s->ntile), dim3(256), 0, 0, msrc, s->mu,
                       s->p, s->q, s->p_pad, 0);
    AQ_HIPF(hipDeviceSynchronize());
    }
    AQ_HIPF(hipFree(stage));
  }
  AQ_TRYF(aq_upload_padded(s->theta, pr->theta_vb, s->p, s->p_pad));
  AQ_TRYF(aq_upload_padded(s->sig2_theta, pr->sig2_theta_vb, s->p, s->p_pad));
  AQ_TRYF(aq_upload_padded(s->eta_h, pr->eta, s->q, s->q_pad));
  AQ_TRYF(aq_upload_padded(s->kappa_h, pr->kappa, s->q, s->q_pad));
  AQ_TRYF(aq_upload_padded(s->n0, pr->n0, s->q, s->q_pad));
  AQ_TRYF(aq_upload_padded(s->zeta, pr->zeta_vb, s->q, s->q_pad));
  AQ_TRYF(aq_upload_padded(s->tau, pr->tau_vb, s->q, s->q_pad));
  AQ_TRYF(aq_upload_padded(s->sig2b, pr->sig2_beta_vb, s->q, s->q_pad));
  {
    std::vector<double> nobs(s->q_pad, 0.0);
    for (int k = 0; k < s->q; k++) {                          // colSums(mis_pat), R/update_vb.R:132
      double cnt = 0;
      for (int i = 0; i < s->n; i++) cnt += (Yh[(size_t)i + (size_t)s->n * k] == Yh[(size_t)i + (size_t)s->n * k]) ? 1.0 : 0.0;
      nobs[k] = cnt;
    }
    AQ_HIPF(hipMemcpy(s->nobs, nobs.data(), nobs.size() * sizeof(double), hipMemcpyHostToDevice));
    // padded traits need valid constants for the init-mode core kernel (coef/inv2s/cst unused there)
    std::vector<double> ones(s->q_pad, 1.0);
    AQ_HIPF(hipMemcpy(s->inv2s, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    if (s->q_pad > s->q) {
      AQ_HIPF(hipMemcpy(s->sig2b + s->q, ones.data(), (size_t)(s->q_pad - s->q) * sizeof(double), hipMemcpyHostToDevice));
      AQ_HIPF(hipMemcpy(s->tau + s->q, ones.data(), (size_t)(s->q_pad - s->q) * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  {
    AqScalars h;
    std::memset(&h, 0, sizeof(h));
    h.sig02_inv = pr->sig02_inv_vb;
    h.lentz_mask[0] = h.lentz_mask[1] = ~0ull;
    AQ_HIPF(hipMemcpy(s->sc, &h, sizeof(h), hipMemcpyHostToDevice));
  }

  // ---- host loop state, R/atlasqtl_global_local_core.R:71-123 ----
  if (!s->has_anneal) {
    s->annealing = false; s->c = s->c_s = 1.0; s->it_init = 1;
  } else {
    s->annealing = true;
    s->ladder = aq_ladder(s->anneal);
    s->c = s->ladder[0]; s->c_s = s->c;            // anneal_scale <- TRUE
    s->it_init = (int)std::llround(s->anneal[2]);
  }
  if (s->thinned) { s->times_conv_sched = {1, 5, 10, 50}; s->batch_conv_sched = {1, 10, 25, 50}; }
  else { s->times_conv_sched = {1}; s->batch_conv_sched = {1}; }
  s->ind_batch_conv = (int)s->batch_conv_sched.size() + 1;
  s->batch_conv = 1;
  s->sig2_zeta = 1.0 / (s->c * ((double)s->p + s->t02_inv));                       // update_sig2_c0_vb_(p, t02, c), :105
  s->vec_sum_log_det_zeta = -(double)s->q_total * (std::log(s->t02) + std::log((double)s->p + s->t02_inv));   // :107
  s->phase = 0;
  *out = s;
  return AQ_OK;
}

extern "C" void aq_vb_destroy(aq_vb_handle h) { aq_free_all(h); }

extern "C" double *aq_vb_reduce_ptr(aq_vb_handle h, int32_t which) {
  if (!h) return nullptr;
  return which == 0 ? h->red : h->ered;
}

// part A of a sweep: S1-S11 + local reductions into the all-reduce payload
static int aq_launch_prepass(aq_vb *s, double c, int do_H) {
  AqPrepass v;
  v.theta = s->theta; v.zeta = s->zeta; v.gam = s->gam; v.Aarr = s->Aarr; v.Barr = s->Barr; v.rowA = s->rowA;
  v.colApart = s->colApart; v.Hpart = s->Hpart; v.p = s->p; v.q = s->q; v.p_pad = s->p_pad; v.q_pad = s->q_pad;
  v.rows_per_chunk = s->rows_per_chunk; v.sqrt_c = std::sqrt(c);
  v.c_is_one = std::fabs(c - 1.0) < 1.5e-8 ? 1 : 0;   // isTRUE(all.equal(c, 1)), R/update_vb.R:219
  v.do_H = do_H;
  v.write_AB = s->fused ? 0 : 1;
  hipLaunchKernelGGL(aq_k_prepass, dim3(s->nHchunk, s->ntile), dim3(256), 0, 0, v);
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

static int aq_sweep_part_a(aq_vb *s) {
  AqQvec qv = aq_qvec(s);
  if (!s->pre_done && !s->fused) AQ_TRY(aq_launch_prepass(s, s->c, 0));
  s->pre_done = false;
  hipLaunchKernelGGL(aq_k_qpre, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, qv, s->sc, s->c);
  AQ_TRY(aq_launch_core(s, 0, s->c));
  hipLaunchKernelGGL(aq_k_reduce_rows, dim3((s->p_pad + 63) / 64), dim3(256), 0, 0, s->fused ? (const double *)nullptr : s->rowA, s->rowGB,
                     s->red, s->ntile, s->p_pad, s->WPT);
  hipLaunchKernelGGL(aq_k_reduce_q_scalars, dim3(1), dim3(1024), 0, 0, qv, s->red + s->p_pad);
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

// part B: S12-S20 on the all-reduced sums, then the ladder step (host scalars)
static int aq_sweep_part_b(aq_vb *s) {
  AqQvec qv = aq_qvec(s);
  AqPvec pv = aq_pvec(s);
  int ann = (s->annealing) ? 1 : 0;   // annealing & anneal_scale, :244
  hipLaunchKernelGGL(aq_k_take_reduced_scalars, dim3(1), dim3(1), 0, 0, s->sc, s->red + s->p_pad);
  if (s->scheme == 1) {   // global-only core: R/atlasqtl_global_core.R:238-256
    hipLaunchKernelGGL(aq_k_pvec_global, dim3(s->pblk), dim3(256), 0, 0, pv, s->sc, s->c, s->q_total);
    hipLaunchKernelGGL(aq_k_scalars_post_global, dim3(1), dim3(1024), 0, 0, pv, s->sc, s->c_s, s->pblk);
  } else {
    hipLaunchKernelGGL(aq_k_reset_lentz, dim3(1), dim3(1), 0, 0, s->sc);
    hipLaunchKernelGGL(aq_k_pvec_L, dim3(s->pblk), dim3(256), 0, 0, pv, s->sc, s->c_s, ann);
    hipLaunchKernelGGL(aq_k_pvec_finish, dim3(s->pblk), dim3(256), 0, 0, pv, s->sc, s->c, s->c_s, ann, s->q_total);
    hipLaunchKernelGGL(aq_k_scalars_post, dim3(1), dim3(1024), 0, 0, pv, s->sc, s->c_s, s->pblk);
  }
  hipLaunchKernelGGL(aq_k_qpost, dim3((s->q_pad + 255) / 256), dim3(256), 0, 0, qv, s->sc, s->c, s->sig2_zeta, s->t02_inv);
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

static int aq_elbo_local(aq_vb *s) {
  AqQvec qv = aq_qvec(s);
  AqPvec pv = aq_pvec(s);
  // the pre-pass of the NEXT sweep (same refreshed theta + zeta, c = 1 here) also yields the p x q ELBO part
  AQ_TRY(aq_launch_prepass(s, s->c, 1));
  s->pre_done = !s->fused;
  if (s->scheme == 1) hipLaunchKernelGGL(aq_k_elbo_C_global, dim3(1), dim3(1), 0, 0, pv, s->sc);
  else hipLaunchKernelGGL(aq_k_elbo_C, dim3(1), dim3(1024), 0, 0, pv, s->sc);
  hipLaunchKernelGGL(aq_k_elbo_q, dim3(1), dim3(1024), 0, 0, qv, s->sc, s->Hpart, s->ntile * s->nHchunk, s->ered);
  AQ_HIP(hipGetLastError());
  return AQ_OK;
}

// Bounded waits inside the sweep kernels (a chained segment waiting for its predecessor, a sample part waiting for its
// partners' partial S) raise errflag when they expire: the results of that launch are invalid.  Polled wherever results
// leave the library: ELBO evaluation, end of a run, status, state and result getters.
static int aq_check_chain_error(aq_vb *s) {
  if (!s->errflag || (s->chain <= 1 && s->misC <= 1 && s->laC <= 1 && !s->errflag_forced)) return AQ_OK;
  int f = 0;
  AQ_HIP(hipMemcpy(&f, s->errflag, sizeof(int), hipMemcpyDeviceToHost));
  if (f != 0) {
    s->failed = true;
    s->fail_code = AQ_ERR_DEVICE;
    s->fail_msg = (s->misC > 1 || s->laC > 1) ? "core sweep: a bounded wait on a partner workgroup's partial sums expired (results invalid)"
                              : "chained core sweep: a bounded wait on a tile's previous SNP segment expired (results invalid; set AQ_CHAIN=0)";
    return aq_fail(AQ_ERR_DEVICE, s->fail_msg);
  }
  return AQ_OK;
}

static int aq_elbo_finish(aq_vb *s, double *lb) {
  AqElboConst k;
  k.nu_h = s->nu; k.rho_h = s->rho; k.A2_inv = s->A2_inv; k.t02_inv = s->t02_inv;
  k.vec_sum_log_det_zeta = s->vec_sum_log_det_zeta; k.sig2_zeta = s->sig2_zeta;
  k.p = (double)s->p; k.q_total = (double)s->q_total;
  k.global_only = s->scheme == 1 ? 1 : 0;
  hipLaunchKernelGGL(aq_k_elbo_final, dim3(1), dim3(1), 0, 0, s->sc, s->ered, k);
  AQ_HIP(hipGetLastError());
  AqScalars h;
  AQ_HIP(hipMemcpy(&h, s->sc, sizeof(h), hipMemcpyDeviceToHost));
  AQ_TRY(aq_check_chain_error(s));
  *lb = h.elbo;
  return AQ_OK;
}

static bool aq_all_equal_1(double c) { return std::fabs(c - 1.0) < 1.5e-8; }

// One step of the state machine.  stop_after_sweeps < 0: unlimited.
static int aq_advance_impl(aq_vb *s, int *sweeps_budget) {
  if (s->failed) return -aq_fail(s->fail_code, "handle is in a failed state: " + s->fail_msg);
  AQ_HIP(hipSetDevice(s->device));
  for (;;) {
    switch (s->phase) {
      case 0: {   // initial residual R = Y - X beta_vb (:112-115 in n-space) and the initial column sums
        AQ_TRY(aq_launch_core(s, 1, 1.0));
        AQ_HIP(hipMemsetAsync(s->red, 0, (size_t)aq_vb_reduce_len(s->p) * sizeof(double), 0));
        AqQvec qv = aq_qvec(s);
        hipLaunchKernelGGL(aq_k_reduce_q_scalars, dim3(1), dim3(1024), 0, 0, qv, s->red + s->p_pad);
        AQ_HIP(hipGetLastError());
        s->phase = 1;
        return AQ_VB_NEED_ALLREDUCE_MAIN;
      }
      case 1:
        hipLaunchKernelGGL(aq_k_take_reduced_scalars, dim3(1), dim3(1), 0, 0, s->sc, s->red + s->p_pad);
        s->phase = 2;
        break;
      case 2:
        if (s->converged || s->it >= s->maxit) return AQ_VB_DONE;                  // :125
        if (sweeps_budget) {
          if (*sweeps_budget == 0) return AQ_VB_DONE;
          (*sweeps_budget)--;
        }
        s->lb_old = s->lb_new;                                                     // :127
        s->it += 1;
        AQ_TRY(aq_sweep_part_a(s));
        s->phase = 3;
        return AQ_VB_NEED_ALLREDUCE_MAIN;
      case 3: {
        AQ_TRY(aq_sweep_part_b(s));
        if (s->annealing) {                                                        // :318-337
          s->sig2_zeta = s->c * s->sig2_zeta;
          s->c = (s->it < (int)s->ladder.size()) ? s->ladder[s->it] : 1.0;         // ladder[it + 1], 1-based
          s->c_s = s->c;
          s->sig2_zeta = s->sig2_zeta / s->c;
          if (aq_all_equal_1(s->c)) s->annealing = false;
          s->phase = 2;
          break;
        }
        bool eval = (s->it <= s->it_init + 1) || (s->it % s->batch_conv == 0) || (s->it % s->batch_conv == 1);   // :342
        if (!eval) {
          s->phase = 2;
          break;
        }
        AQ_TRY(aq_elbo_local(s));
        s->phase = 4;
        return AQ_VB_NEED_ALLREDUCE_ELBO;
      }
      case 4: {
        double lb;
        AQ_TRY(aq_elbo_finish(s, &lb));
        s->lb_new = lb;
        s->trace_it.push_back(s->it);
        s->trace_lb.push_back(lb);
        const double eps = std::sqrt(std::numeric_limits<double>::epsilon());      // :85
        if (s->debug && lb + eps < s->lb_old) {                                    // :359-360
          s->failed = true;
          s->fail_code = AQ_ERR_NUMERIC;
          char buf[256];
          std::snprintf(buf, sizeof(buf), "ELBO not increasing monotonically. Exit. (it=%d, lb_old=%.17g, lb_new=%.17g)", s->it,
                        s->lb_old, lb);
          s->fail_msg = buf;
          return -aq_fail(AQ_ERR_NUMERIC, buf);
        }
        double diff = std::fabs(lb - s->lb_old);                                   // :362
        int sum_exceed = 0;
        for (double t : s->times_conv_sched) sum_exceed += (diff > t * s->tol) ? 1 : 0;   // :364
        if (sum_exceed == 0) {
          s->converged = true;
        } else if (s->ind_batch_conv > sum_exceed) {
          s->ind_batch_conv = sum_exceed;
          s->batch_conv = s->batch_conv_sched[sum_exceed - 1];
        }
        s->phase = 2;
        break;
      }
      default:
        return -aq_fail(AQ_ERR_ARG, "corrupt state");
    }
  }
}

extern "C" int aq_vb_advance(aq_vb_handle h) {
  if (!h) return -aq_fail(AQ_ERR_ARG, "NULL handle");
  int rc = aq_advance_impl(h, h->budget >= 0 ? &h->budget : nullptr);
  return rc;
}
extern "C" int aq_vb_set_sweep_budget(aq_vb_handle h, int32_t sweeps) {
  if (!h) return aq_fail(AQ_ERR_ARG, "NULL handle");
  h->budget = sweeps < 0 ? -1 : sweeps;
  return AQ_OK;
}

static int aq_run_impl(aq_vb *s, int *budget) {
  if (s->world != 1) return aq_fail(AQ_ERR_ARG, "aq_vb_run: world_size != 1 needs the aq_vb_advance protocol");
  for (;;) {
    int rc = aq_advance_impl(s, budget);
    if (rc < 0) return -rc;
    if (rc == AQ_VB_DONE) break;
  }
  AQ_HIP(hipDeviceSynchronize());
  return aq_check_chain_error(s);
}
extern "C" int aq_vb_run(aq_vb_handle h) {
  if (!h) return aq_fail(AQ_ERR_ARG, "NULL handle");
  return aq_run_impl(h, nullptr);
}
extern "C" int aq_vb_run_sweeps(aq_vb_handle h, int32_t max_sweeps) {
  if (!h) return aq_fail(AQ_ERR_ARG, "NULL handle");
  int budget = max_sweeps;
  return aq_run_impl(h, &budget);
}

static void aq_resolve_events(aq_vb *s) {
  for (auto &e : s->ev) {
    float ms = 0.f;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
      s->core_ms_acc += ms;
      s->core_launches++;
    }
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  s->ev.clear();
}

// ------------------------------------------------------------ post-processing ----
extern "C" int aq_assign_bfdr(const double *mat_ppi, double *mat_fdr, int64_t len, int32_t device) {
  if (!mat_ppi || !mat_fdr || len < 0) return aq_fail(AQ_ERR_ARG, "aq_assign_bfdr: bad argument");
  AQ_TRY(aq_need_device(device));
  if (len == 0) return AQ_OK;
  double *din = nullptr, *dout = nullptr;
  AQ_HIP(hipMalloc((void **)&din, (size_t)len * sizeof(double)));
  AQ_HIP(hipMalloc((void **)&dout, (size_t)len * sizeof(double)));
  AQ_HIP(hipMemcpy(din, mat_ppi, (size_t)len * sizeof(double), hipMemcpyHostToDevice));
  int rc = aq_bfdr_device(din, dout, len);
  if (rc == AQ_OK && hipMemcpy(mat_fdr, dout, (size_t)len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    rc = aq_fail(AQ_ERR_DEVICE, "aq_assign_bfdr: copy back failed");
  hipFree(din);
  hipFree(dout);
  return rc;
}

// d_m: p x q column-major PPIs on the device (overwritten by the FDR matrix when fdr_adjust)
static int aq_hotspot_common(double *d_m, int p, int q, double thres, int fdr_adjust, int64_t *rs_thres, int64_t *nb_pairwise) {
  int lt = 0;
  if (fdr_adjust) {
    double *d_f = nullptr;
    AQ_HIP(hipMalloc((void **)&d_f, (size_t)p * q * sizeof(double)));
    int rc = aq_bfdr_device(d_m, d_f, (int64_t)p * q);
    if (rc == AQ_OK) {
      hipError_t e = hipMemcpy(d_m, d_f, (size_t)p * q * sizeof(double), hipMemcpyDeviceToDevice);
      if (e != hipSuccess) rc = aq_fail(AQ_ERR_DEVICE, "aq_hotspot_sizes: device copy failed");
    }
    hipFree(d_f);
    if (rc != AQ_OK) return rc;
    lt = 1;                                                      // rowSums(mat_fdr < thres), R/summarise_output.R:100
  }
  int64_t *d_rs = nullptr;
  AQ_HIP(hipMalloc((void **)&d_rs, (size_t)p * sizeof(int64_t)));
  int rc = aq_row_count_device(d_m, d_rs, p, q, thres, lt);    // rowSums(gam_vb > thres), :103
  std::vector<int64_t> rs(p);
  if (rc == AQ_OK && hipMemcpy(rs.data(), d_rs, (size_t)p * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
    rc = aq_fail(AQ_ERR_DEVICE, "aq_hotspot_sizes: copy back failed");
  hipFree(d_rs);
  if (rc != AQ_OK) return rc;
  int64_t tot = 0;
  for (int j = 0; j < p; j++) { tot += rs[j]; if (rs_thres) rs_thres[j] = rs[j]; }
  if (nb_pairwise) *nb_pairwise = tot;                           // sum(gam_vb > thres), :102
  return AQ_OK;
}

extern "C" int aq_hotspot_sizes(const double *mat_ppi, int32_t p, int32_t q, double thres, int32_t fdr_adjust,
                                int64_t *rs_thres, int64_t *nb_pairwise, int32_t device) {
  if (!mat_ppi || p <= 0 || q <= 0) return aq_fail(AQ_ERR_ARG, "aq_hotspot_sizes: bad argument");
  AQ_TRY(aq_need_device(device));
  double *d_m = nullptr;
  AQ_HIP(hipMalloc((void **)&d_m, (size_t)p * q * sizeof(double)));
  AQ_HIP(hipMemcpy(d_m, mat_ppi, (size_t)p * q * sizeof(double), hipMemcpyHostToDevice));
  int rc = aq_hotspot_common(d_m, p, q, thres, fdr_adjust, rs_thres, nb_pairwise);
  hipFree(d_m);
  return rc;
}

extern "C" int aq_vb_hotspot_sizes(aq_vb_handle s, double thres, int32_t fdr_adjust, int64_t *rs_thres, int64_t *nb_pairwise) {
  if (!s) return aq_fail(AQ_ERR_ARG, "NULL handle");
  AQ_HIP(hipSetDevice(s->device));
  double *d_m = nullptr;
  AQ_HIP(hipMalloc((void **)&d_m, (size_t)s->p * s->q * sizeof(double)));
  hipLaunchKernelGGL(aq_k_colmajor_from_tile, dim3((s->p_pad + 63) / 64, s->ntile), dim3(256), 0, 0, s->gam, (const double *)nullptr,
                     d_m, s->p, s->q, s->p_pad);
  int rc = aq_hotspot_common(d_m, s->p, s->q, thres, fdr_adjust, rs_thres, nb_pairwise);
  hipFree(d_m);
  return rc;
}

// Bayesian FDR under trait sharding (aq_postproc.hip): the caller bisects over a PPI cutoff, all-reducing the five numbers
// of aq_vb_bfdr_query over the ranks at every step (atlasqtl_amd/core.py::VbRun.hotspot_sizes).
extern "C" int aq_vb_bfdr_begin(aq_vb_handle s) {
  if (!s) return aq_fail(AQ_ERR_ARG, "NULL handle");
  AQ_HIP(hipSetDevice(s->device));
  if (s->bf) { aq_shard_free(s->bf); s->bf = nullptr; }
  double *d_m = nullptr;
  AQ_HIP(hipMalloc((void **)&d_m, (size_t)s->p * s->q * sizeof(double)));
  hipLaunchKernelGGL(aq_k_colmajor_from_tile, dim3((s->p_pad + 63) / 64, s->ntile), dim3(256), 0, 0, s->gam, (const double *)nullptr,
                     d_m, s->p, s->q, s->p_pad);
  int rc = aq_shard_sort(d_m, (int64_t)s->p * s->q, &s->bf);
  hipFree(d_m);
  return rc;
}
extern "C" int aq_vb_bfdr_query(aq_vb_handle s, double c, double *out5) {
  if (!s || !out5 || !s->bf) return aq_fail(AQ_ERR_ARG, "aq_vb_bfdr_query: call aq_vb_bfdr_begin first");
  AQ_HIP(hipSetDevice(s->device));
  return aq_shard_query(s->bf, c, out5);
}
extern "C" int aq_vb_bfdr_rows(aq_vb_handle s, int64_t upto, int64_t tie_first, int64_t take, int64_t *rs) {
  if (!s || !rs || !s->bf) return aq_fail(AQ_ERR_ARG, "aq_vb_bfdr_rows: call aq_vb_bfdr_begin first");
  if (upto < 0 || take < 0 || tie_first < 0 || upto > (int64_t)s->p * s->q || tie_first + take > (int64_t)s->p * s->q)
    return aq_fail(AQ_ERR_ARG, "aq_vb_bfdr_rows: positions out of range");
  AQ_HIP(hipSetDevice(s->device));
  return aq_shard_rows(s->bf, upto, tie_first, take, s->p, rs);
}
extern "C" void aq_vb_bfdr_end(aq_vb_handle s) {
  if (s && s->bf) { hipSetDevice(s->device); aq_shard_free(s->bf); s->bf = nullptr; }
}

// ------------------------------------------------------ checkpoint / resume ----
// The reference's checkpoint_ (R/utils.R:571-611) only writes outputs; it cannot resume.  Here the complete loop state
// between two sweeps is one flat blob: header, host-side loop scalars, ELBO trace, then the device arrays the next
// sweep reads (gam, mu, the incrementally updated residual, p- and q-vectors, column sums, AqScalars).  A, b and the row
// sums of the pre-pass are not stored: the next sweep recomputes them from theta and zeta (same kernel, same bits).
struct AqStateHeader {
  uint64_t magic;        // "AQVBST02"
  int32_t n, p, q, q_total, p_pad, q_pad, n_pad, core_kernel;
  int32_t it, converged, annealing, ind_batch_conv, batch_conv, failed, n_trace, has_missing;
  int32_t trait_offset, scheme_df;  // which trait shard of a q-sharded run the state belongs to; scheme + 16 df
  double c, c_s, sig2_zeta, lb_new, lb_old;
};
static const uint64_t AQ_STATE_MAGIC = 0x32305453425651ull | ((uint64_t)'A' << 56);

struct AqStateSeg { void *ptr; size_t bytes; };
static std::vector<AqStateSeg> aq_state_segments(aq_vb *s) {
  const size_t pq = (size_t)s->ntile * s->p_pad * 16 * sizeof(double);
  const size_t P = (size_t)s->p_pad * sizeof(double), Q = (size_t)s->q_pad * sizeof(double);
  std::vector<AqStateSeg> v = {
      {s->gam, pq}, {s->mu, pq}, {s->R, (size_t)s->ntile * s->n_pad * 16 * sizeof(double)},
      {s->theta, P}, {s->sig2_theta, P}, {s->L, P}, {s->lam2_inv, P}, {s->Q, P},
      {s->zeta, Q}, {s->tau, Q}, {s->sig2b, Q}, {s->log_tau, Q}, {s->eta_vb, Q}, {s->kappa_vb, Q},
      {s->coef, Q}, {s->inv2s, Q}, {s->cst, Q}, {s->sums, 6 * Q}, {s->sc, sizeof(AqScalars)}};
  return v;
}
static int aq_core_kernel_id(const aq_vb *s) { return s->use_mis ? 3 : s->use_tw ? 2 : 0; }

extern "C" int64_t aq_vb_state_bytes(aq_vb_handle s) {
  if (!s) return -1;
  size_t tot = sizeof(AqStateHeader) + s->trace_it.size() * (sizeof(int32_t) + sizeof(double));
  for (auto &g : aq_state_segments(s)) tot += g.bytes;
  return (int64_t)tot;
}

extern "C" int aq_vb_get_state(aq_vb_handle s, void *buf, int64_t cap) {
  if (!s || !buf) return aq_fail(AQ_ERR_ARG, "NULL argument");
  if (s->phase != 2) return aq_fail(AQ_ERR_ARG, "aq_vb_get_state: only between sweeps (after aq_vb_run / aq_vb_run_sweeps returned)");
  if (cap < aq_vb_state_bytes(s)) return aq_fail(AQ_ERR_ARG, "aq_vb_get_state: buffer too small");
  AQ_HIP(hipSetDevice(s->device));
  AQ_HIP(hipDeviceSynchronize());
  AQ_TRY(aq_check_chain_error(s));
  AqStateHeader h;
  std::memset(&h, 0, sizeof(h));
  h.magic = AQ_STATE_MAGIC;
  h.n = s->n; h.p = s->p; h.q = s->q; h.q_total = s->q_total; h.p_pad = s->p_pad; h.q_pad = s->q_pad; h.n_pad = s->n_pad;
  h.core_kernel = aq_core_kernel_id(s); h.trait_offset = s->trait_offset; h.scheme_df = s->scheme + 16 * s->df;
  h.it = s->it; h.converged = s->converged; h.annealing = s->annealing; h.ind_batch_conv = s->ind_batch_conv;
  h.batch_conv = s->batch_conv; h.failed = s->failed; h.n_trace = (int32_t)s->trace_it.size(); h.has_missing = s->has_missing;
  h.c = s->c; h.c_s = s->c_s; h.sig2_zeta = s->sig2_zeta; h.lb_new = s->lb_new; h.lb_old = s->lb_old;
  char *o = (char *)buf;
  std::memcpy(o, &h, sizeof(h)); o += sizeof(h);
  for (int i = 0; i < h.n_trace; i++) { int32_t v = s->trace_it[i]; std::memcpy(o, &v, sizeof(v)); o += sizeof(v); }
  for (int i = 0; i < h.n_trace; i++) { double v = s->trace_lb[i]; std::memcpy(o, &v, sizeof(v)); o += sizeof(v); }
  for (auto &g : aq_state_segments(s)) {
    AQ_HIP(hipMemcpy(o, g.ptr, g.bytes, hipMemcpyDeviceToHost));
    o += g.bytes;
  }
  return AQ_OK;
}

extern "C" int aq_vb_set_state(aq_vb_handle s, const void *buf, int64_t len) {
  if (!s || !buf) return aq_fail(AQ_ERR_ARG, "NULL argument");
  if (len < (int64_t)sizeof(AqStateHeader)) return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: truncated state");
  AqStateHeader h;
  const char *o = (const char *)buf;
  std::memcpy(&h, o, sizeof(h)); o += sizeof(h);
  if (h.magic != AQ_STATE_MAGIC) return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: not an atlasqtl-hip state blob");
  if (h.n != s->n || h.p != s->p || h.q != s->q || h.q_total != s->q_total || h.has_missing != (int)s->has_missing)
    return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: the state was saved for a different problem shape");
  if (h.core_kernel != aq_core_kernel_id(s))
    return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: the state was saved by core kernel " + std::to_string(h.core_kernel) + ", this handle runs kernel " +
                                   std::to_string(aq_core_kernel_id(s)) + " (0 look-ahead MFMA, 2 generic, 3 masked two-barrier): create the handle with AQ_KERNEL / "
                                   "AQ_GK_MAX_GB set as for the run that saved it");
  if (h.p_pad != s->p_pad || h.q_pad != s->q_pad || h.n_pad != s->n_pad)
    return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: same problem, different kernel geometry (padding " + std::to_string(h.n_pad) + " / " + std::to_string(h.q_pad) +
                                   " saved, " + std::to_string(s->n_pad) + " / " + std::to_string(s->q_pad) + " here): the launch plan depends on the "
                                   "device's CU count and on AQ_TT / AQ_LA_C / AQ_NT3; resume with the settings of the run that saved the state");
  if (h.trait_offset != s->trait_offset)
    return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: the state belongs to another trait shard (trait_offset differs)");
  if (h.scheme_df != s->scheme + 16 * s->df)
    return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: the state was saved under another scheme (global-local / global-only) or df");
  size_t need = sizeof(h) + (size_t)h.n_trace * (sizeof(int32_t) + sizeof(double));
  for (auto &g : aq_state_segments(s)) need += g.bytes;
  if (h.n_trace < 0 || (int64_t)need != len) return aq_fail(AQ_ERR_ARG, "aq_vb_set_state: state size mismatch");
  AQ_HIP(hipSetDevice(s->device));
  AQ_HIP(hipDeviceSynchronize());
  s->trace_it.resize(h.n_trace);
  s->trace_lb.resize(h.n_trace);
  for (int i = 0; i < h.n_trace; i++) { int32_t v; std::memcpy(&v, o, sizeof(v)); o += sizeof(v); s->trace_it[i] = v; }
  for (int i = 0; i < h.n_trace; i++) { double v; std::memcpy(&v, o, sizeof(v)); o += sizeof(v); s->trace_lb[i] = v; }
  for (auto &g : aq_state_segments(s)) {
    AQ_HIP(hipMemcpy(g.ptr, o, g.bytes, hipMemcpyHostToDevice));
    o += g.bytes;
  }
  s->it = h.it; s->converged = h.converged != 0; s->annealing = h.annealing != 0; s->ind_batch_conv = h.ind_batch_conv;
  s->batch_conv = h.batch_conv; s->failed = h.failed != 0;
  s->c = h.c; s->c_s = h.c_s; s->sig2_zeta = h.sig2_zeta; s->lb_new = h.lb_new; s->lb_old = h.lb_old;
  s->pre_done = false;   // the next sweep recomputes the pre-pass from the restored theta / zeta
  s->phase = 2;
  return AQ_OK;
}

extern "C" int aq_vb_get_status(aq_vb_handle s, aq_vb_status *st) {
  if (!s || !st) return aq_fail(AQ_ERR_ARG, "NULL argument");
  AQ_HIP(hipSetDevice(s->device));
  aq_resolve_events(s);
  AQ_TRY(aq_check_chain_error(s));
  AqScalars h;
  AQ_HIP(hipMemcpy(&h, s->sc, sizeof(h), hipMemcpyDeviceToHost));
  st->it = s->it;
  st->converged = s->converged ? 1 : 0;
  st->lb_opt = s->lb_new;
  st->diff_lb = std::fabs(s->lb_new - s->lb_old);
  st->c = s->c;
  st->annealing = s->annealing ? 1 : 0;
  st->n_elbo = (int)s->trace_it.size();
  st->core_ms = s->core_ms_acc;
  st->core_launches = s->core_launches;
  st->sig02_inv_vb = h.sig02_inv;
  st->sig2_inv_vb = h.sig2_inv;
  st->lentz_iters = h.lentz_iters;
  st->core_kernel = aq_core_kernel_id(s);
  st->split_parts = s->use_la ? s->laC : s->use_mis ? s->misC : 1;
  st->tiles_per_group = s->use_la ? s->TT : 1;
  st->chain_segments = s->chain > 1 ? s->chain : 0;
  return AQ_OK;
}

extern "C" int32_t aq_vb_get_overrides(aq_vb_handle s, char *buf, int32_t cap) {
  if (!s) return -1;
  const int32_t n = (int32_t)s->overrides.size();
  if (buf && cap > 0) {
    const int32_t m = n < cap - 1 ? n : cap - 1;
    std::memcpy(buf, s->overrides.data(), (size_t)m);
    buf[m] = 0;
  }
  return n;
}

extern "C" int32_t aq_vb_get_elbo_trace(aq_vb_handle s, int32_t *it_out, double *lb_out, int32_t cap) {
  if (!s) return 0;
  int n = (int)s->trace_it.size();
  for (int i = 0; i < n && i < cap; i++) {
    if (it_out) it_out[i] = s->trace_it[i];
    if (lb_out) lb_out[i] = s->trace_lb[i];
  }
  return n;
}

extern "C" int aq_vb_get_result(aq_vb_handle s, double *beta_vb, double *gam_vb, double *mu_beta_vb, double *theta_vb,
                                double *zeta_vb, double *lam2_inv_vb, double *sig2_theta_vb, double *tau_vb,
                                double *sig2_beta_vb) {
  if (!s) return aq_fail(AQ_ERR_ARG, "NULL handle");
  AQ_HIP(hipSetDevice(s->device));
  AQ_HIP(hipDeviceSynchronize());
  AQ_TRY(aq_check_chain_error(s));
  size_t pq = (size_t)s->p * s->q;
  if (beta_vb || gam_vb || mu_beta_vb) {
    double *stage;
    AQ_HIP(hipMalloc((void **)&stage, pq * sizeof(double)));
    dim3 grid((s->p_pad + 63) / 64, s->ntile);
    struct { double *dst; const double *src; const double *mul; } jobs[3] = {
        {beta_vb, s->gam, s->mu}, {gam_vb, s->gam, nullptr}, {mu_beta_vb, s->mu, nullptr}};
    for (auto &j : jobs) {
      if (!j.dst) continue;
      hipLaunchKernelGGL(aq_k_colmajor_from_tile, grid, dim3(256), 0, 0, j.src, j.mul, stage, s->p, s->q, s->p_pad);
      hipError_t e = hipMemcpy(j.dst, stage, pq * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess) { hipFree(stage); return aq_fail(AQ_ERR_DEVICE, hipGetErrorString(e)); }
    }
    AQ_HIP(hipFree(stage));
  }
  if (theta_vb) AQ_HIP(hipMemcpy(theta_vb, s->theta, (size_t)s->p * sizeof(double), hipMemcpyDeviceToHost));
  if (zeta_vb) AQ_HIP(hipMemcpy(zeta_vb, s->zeta, (size_t)s->q * sizeof(double), hipMemcpyDeviceToHost));
  if (lam2_inv_vb) AQ_HIP(hipMemcpy(lam2_inv_vb, s->lam2_inv, (size_t)s->p * sizeof(double), hipMemcpyDeviceToHost));
  if (sig2_theta_vb) AQ_HIP(hipMemcpy(sig2_theta_vb, s->sig2_theta, (size_t)s->p * sizeof(double), hipMemcpyDeviceToHost));
  if (tau_vb) AQ_HIP(hipMemcpy(tau_vb, s->tau, (size_t)s->q * sizeof(double), hipMemcpyDeviceToHost));
  if (sig2_beta_vb) AQ_HIP(hipMemcpy(sig2_beta_vb, s->sig2b, (size_t)s->q * sizeof(double), hipMemcpyDeviceToHost));
  return AQ_OK;
}

// The residual the sweep kernel carries in n-space, mis_pat .* (Y - X beta_vb) (= what cp_Y_X - cp_betaX_X of the reference
// encodes, src/coreLoop.cpp:71,81), n x q column-major.  It is only ever updated incrementally (R -= X delta per SNP block and
// sweep), so comparing it with Y - X beta_vb recomputed from the returned beta_vb measures the rounding drift of a whole run.
extern "C" int aq_vb_get_residual(aq_vb_handle s, double *R_out) {
  if (!s || !R_out) return aq_fail(AQ_ERR_ARG, "NULL argument");
  AQ_HIP(hipSetDevice(s->device));
  AQ_HIP(hipDeviceSynchronize());
  AQ_TRY(aq_check_chain_error(s));
  if (s->use_tw && s->WPT > 1) return aq_fail(AQ_ERR_UNSUPPORTED, "aq_vb_get_residual: not for the generic kernel's split layout");
  const size_t nq = (size_t)s->n * s->q;
  double *stage = nullptr;
  AQ_HIP(hipMalloc((void **)&stage, nq * sizeof(double)));
  hipLaunchKernelGGL(aq_k_colmajor_from_tile, dim3((s->n_pad + 63) / 64, s->ntile), dim3(256), 0, 0, s->R, (const double *)nullptr,
                     stage, s->n, s->q, s->n_pad);
  hipError_t e = hipMemcpy(R_out, stage, nq * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(stage);
  if (e != hipSuccess) return aq_fail(AQ_ERR_DEVICE, hipGetErrorString(e));
  return AQ_OK;
}

// ------------------------------------------------- operator-level entries ----
static int aq_gram_common(bool mis, const double *cp_X, const double *const *cp_X_rm, const double *cp_Y_X, double *gam_vb,
                          const double *lP, const double *l1, double log_sig2_inv_vb, const double *log_tau_vb,
                          double *m1_beta, double *cp_betaX_X, double *mu_beta_vb, const double *sig2_beta_vb,
                          const double *tau_vb, const int32_t *shuffled_ind, int32_t n_ind, const int32_t *sample_q,
                          int32_t n_q, double c, int32_t p, int32_t q) {
  if (!cp_X || !cp_Y_X || !gam_vb || !lP || !l1 || !log_tau_vb || !m1_beta || !cp_betaX_X || !mu_beta_vb || !sig2_beta_vb ||
      !tau_vb || (mis && !cp_X_rm))
    return aq_fail(AQ_ERR_ARG, "aq_core_dual_loop: NULL argument");
  if (p < 1 || q < 1 || n_ind < 0 || n_q < 0) return aq_fail(AQ_ERR_ARG, "aq_core_dual_loop: bad sizes");
  if ((n_ind > 0 && !shuffled_ind) || (n_q > 0 && !sample_q)) return aq_fail(AQ_ERR_ARG, "aq_core_dual_loop: NULL index vector");
  for (int i = 0; i < n_ind; i++)
    if (shuffled_ind[i] < 0 || shuffled_ind[i] >= p) return aq_fail(AQ_ERR_ARG, "shuffled_ind out of range [0, p)");
  {
    std::vector<char> seen((size_t)q, 0);
    for (int i = 0; i < n_q; i++) {
      if (sample_q[i] < 0 || sample_q[i] >= q) return aq_fail(AQ_ERR_ARG, "sample_q out of range [0, q)");
      if (seen[sample_q[i]]) return aq_fail(AQ_ERR_ARG, "sample_q holds a repeated trait index");
      seen[sample_q[i]] = 1;
    }
  }
  AQ_TRY(aq_need_device(0));
  if (n_ind == 0 || n_q == 0) return AQ_OK;   // empty index vectors: nothing to do (the reference's loops do not execute)

  size_t pp = (size_t)p * p, pq = (size_t)p * q;
  std::vector<void *> to_free;
  auto cleanup = [&]() { for (void *x : to_free) hipFree(x); };
  auto up = [&](const void *src, size_t bytes, void **dst) -> int {
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return aq_fail(AQ_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    to_free.push_back(*dst);
    e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return aq_fail(AQ_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    return AQ_OK;
  };
#define AQ_UP(src, count, dst)                                            \
  do {                                                                    \
    int rcu_ = up((src), (count) * sizeof(*(src)), (void **)&(dst));      \
    if (rcu_ != AQ_OK) { cleanup(); return rcu_; }                        \
  } while (0)
  AqGramArgs a;
  std::memset(&a, 0, sizeof(a));
  double *d_cpX, *d_cpYX, *d_gam, *d_lP, *d_l1, *d_lt, *d_m1, *d_bx, *d_mu, *d_s2, *d_tau;
  int32_t *d_si, *d_sq;
  AQ_UP(cp_X, pp, d_cpX);
  AQ_UP(cp_Y_X, pq, d_cpYX);
  AQ_UP(gam_vb, pq, d_gam);
  AQ_UP(lP, pq, d_lP);
  AQ_UP(l1, pq, d_l1);
  AQ_UP(log_tau_vb, (size_t)q, d_lt);
  AQ_UP(m1_beta, pq, d_m1);
  AQ_UP(cp_betaX_X, pq, d_bx);
  AQ_UP(mu_beta_vb, pq, d_mu);
  AQ_UP(sig2_beta_vb, mis ? pq : (size_t)q, d_s2);
  AQ_UP(tau_vb, (size_t)q, d_tau);
  AQ_UP(shuffled_ind, (size_t)n_ind, d_si);
  AQ_UP(sample_q, (size_t)n_q, d_sq);
  const double **d_rm_arr = nullptr;
  if (mis) {
    std::vector<const double *> hp((size_t)q, nullptr);
    for (int k = 0; k < q; k++) {
      if (!cp_X_rm[k]) { cleanup(); return aq_fail(AQ_ERR_ARG, "cp_X_rm holds a NULL matrix"); }
      double *dk;
      AQ_UP(cp_X_rm[k], pp, dk);
      hp[k] = dk;
    }
    const double **tmp;
    int rcu = up(hp.data(), (size_t)q * sizeof(const double *), (void **)&tmp);
    if (rcu != AQ_OK) { cleanup(); return rcu; }
    d_rm_arr = tmp;
  }
  a.cp_X = d_cpX; a.cp_X_rm = d_rm_arr; a.cp_Y_X = d_cpYX; a.gam_vb = d_gam; a.log_Phi = d_lP; a.log_1mPhi = d_l1;
  a.log_sig2_inv_vb = log_sig2_inv_vb; a.log_tau_vb = d_lt; a.m1_beta = d_m1; a.cp_betaX_X = d_bx; a.mu_beta_vb = d_mu;
  a.sig2_beta_vb = d_s2; a.tau_vb = d_tau; a.shuffled_ind = d_si; a.n_ind = n_ind; a.sample_q = d_sq; a.n_q = n_q;
  a.c = c; a.p = p; a.q = q;
  int grid = n_q < 2048 ? n_q : 2048;
  if (mis) hipLaunchKernelGGL((aq_gram_loop_kernel<true>), dim3(grid), dim3(256), 0, 0, a);
  else hipLaunchKernelGGL((aq_gram_loop_kernel<false>), dim3(grid), dim3(256), 0, 0, a);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(gam_vb, d_gam, pq * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(m1_beta, d_m1, pq * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cp_betaX_X, d_bx, pq * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(mu_beta_vb, d_mu, pq * sizeof(double), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return aq_fail(AQ_ERR_DEVICE, std::string("aq_core_dual_loop: ") + hipGetErrorString(e));
  return AQ_OK;
#undef AQ_UP
}

extern "C" int aq_core_dual_loop(const double *cp_X, const double *cp_Y_X, double *gam_vb, const double *lP, const double *l1,
                                 double log_sig2_inv_vb, const double *log_tau_vb, double *m1_beta, double *cp_betaX_X,
                                 double *mu_beta_vb, const double *sig2_beta_vb, const double *tau_vb,
                                 const int32_t *shuffled_ind, int32_t n_ind, const int32_t *sample_q, int32_t n_q, double c,
                                 int32_t p, int32_t q) {
  return aq_gram_common(false, cp_X, nullptr, cp_Y_X, gam_vb, lP, l1, log_sig2_inv_vb, log_tau_vb, m1_beta, cp_betaX_X,
                        mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, n_ind, sample_q, n_q, c, p, q);
}
extern "C" int aq_core_dual_mis_loop(const double *cp_X, const double *const *cp_X_rm, const double *cp_Y_X, double *gam_vb,
                                     const double *lP, const double *l1, double log_sig2_inv_vb, const double *log_tau_vb,
                                     double *m1_beta, double *cp_betaX_X, double *mu_beta_vb, const double *sig2_beta_vb,
                                     const double *tau_vb, const int32_t *shuffled_ind, int32_t n_ind,
                                     const int32_t *sample_q, int32_t n_q, double c, int32_t p, int32_t q) {
  return aq_gram_common(true, cp_X, cp_X_rm, cp_Y_X, gam_vb, lP, l1, log_sig2_inv_vb, log_tau_vb, m1_beta, cp_betaX_X,
                        mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, n_ind, sample_q, n_q, c, p, q);
}

// ------------------------------------------------------------- test hooks ----
extern "C" int aq_vb_debug_raise_errflag(aq_vb_handle s) {
  if (!s || !s->errflag) return aq_fail(AQ_ERR_ARG, "NULL handle");
  AQ_HIP(hipSetDevice(s->device));
  int one = 1;
  AQ_HIP(hipMemcpy(s->errflag, &one, sizeof(int), hipMemcpyHostToDevice));
  s->errflag_forced = true;
  return AQ_OK;
}

// one element of the test hook, compiled for host and device from the same header the kernels use
__host__ __device__ static inline bool aq_special_one(int which, double x, double x2, double *out) {
  double a_, b_, c_, d_;
  switch (which) {
    case 0: *out = aq_log_ndtr(x); return true;
    case 1: *out = aq_digamma(x); return true;
    case 2: *out = aq_expint_e1_small(x); return true;
    case 3: *out = aq_gamma_inc_upper(x2, x); return true;
    case 4: *out = aq_sigmoid_neg(x); return true;
    case 5: aq_log_ndtr_pair(x, &a_, &b_); *out = a_; return true;
    case 6: aq_log_ndtr_pair(x, &a_, &b_); *out = b_; return true;
    case 7: aq_probit_terms(x, &a_, &b_, &c_, &d_); *out = c_; return true;
    case 8: aq_probit_terms(x, &a_, &b_, &c_, &d_); *out = d_; return true;
    case 9: *out = aq_erfcx_pos(x); return true;
    case 10: aq_probit_A_imr(x, &a_, &b_, &c_, &d_); *out = a_; return true;
    case 11: aq_probit_A_imr(x, &a_, &b_, &c_, &d_); *out = b_; return true;
    case 12: aq_probit_A_imr(x, &a_, &b_, &c_, &d_); *out = c_; return true;
    case 13: *out = aq_sigmoid_neg_fast(x); return true;
    // compute_integral_hs_(alpha = df, beta = L df, m, n, Q(L)) for the horseshoe's df = 5 (14: m = n = 3, 15: m = 3, n = 2) and
    // df = 7 (16: m = n = 4, 17: m = 4, n = 3); x = L, x2 = Q_approx(L)
    case 14: *out = aq_hs_integral(5.0, 5.0 * x, 3, 3, x2); return true;
    case 15: *out = aq_hs_integral(5.0, 5.0 * x, 3, 2, x2); return true;
    case 16: *out = aq_hs_integral(7.0, 7.0 * x, 4, 4, x2); return true;
    case 17: *out = aq_hs_integral(7.0, 7.0 * x, 4, 3, x2); return true;
    // the table-driven probit terms of the sweep kernel's helper wave (aq_probit_tab.h): A, imr1, imr0
    // update_annealed_lam2_inv_vb_ for df = 3, 5, 7 (R/update_vb.R:76-81): x = L_vb, x2 = c
    case 21: *out = aq_annealed_lam2_inv_df(x, x2, 3.0); return true;
    case 22: *out = aq_annealed_lam2_inv_df(x, x2, 5.0); return true;
    case 23: *out = aq_annealed_lam2_inv_df(x, x2, 7.0); return true;
    case 18: aq_probit_A_imr_tab(x, aq_pt_table(), &a_, &b_, &c_); *out = a_; return true;
    case 19: aq_probit_A_imr_tab(x, aq_pt_table(), &a_, &b_, &c_); *out = b_; return true;
    case 20: aq_probit_A_imr_tab(x, aq_pt_table(), &a_, &b_, &c_); *out = c_; return true;
    // log Phi(x), log(1 - Phi(x)) from the tables, as the ELBO pass takes them (aq_log_ndtr_pair_tab)
    case 24: aq_log_ndtr_pair_tab(x, aq_pt_table(), aq_ptn_table(), &a_, &b_); *out = a_; return true;
    case 25: aq_log_ndtr_pair_tab(x, aq_pt_table(), aq_ptn_table(), &a_, &b_); *out = b_; return true;
    default: return false;
  }
}

extern "C" int aq_special_eval(int32_t which, const double *x, const double *x2, double *out, int64_t len) {
  if (!x || !out || len < 0) return aq_fail(AQ_ERR_ARG, "aq_special_eval: bad argument");
  if ((which == 3 || (which >= 14 && which <= 17) || (which >= 21 && which <= 23)) && !x2) return aq_fail(AQ_ERR_ARG, "aq_special_eval: x2 required");
  for (int64_t i = 0; i < len; i++)
    if (!aq_special_one(which, x[i], x2 ? x2[i] : 0.0, &out[i])) return aq_fail(AQ_ERR_ARG, "aq_special_eval: unknown function id");
  return AQ_OK;
}

__global__ void aq_k_special_eval(int which, const double *x, const double *x2, double *out, long long len) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) aq_special_one(which, x[i], x2 ? x2[i] : 0.0, &out[i]);
}

extern "C" int aq_special_eval_device(int32_t which, const double *x, const double *x2, double *out, int64_t len, int32_t device) {
  if (!x || !out || len < 0) return aq_fail(AQ_ERR_ARG, "aq_special_eval_device: bad argument");
  if (which < 0 || which > 25) return aq_fail(AQ_ERR_ARG, "aq_special_eval_device: unknown function id");
  if ((which == 3 || (which >= 14 && which <= 17) || (which >= 21 && which <= 23)) && !x2) return aq_fail(AQ_ERR_ARG, "aq_special_eval_device: x2 required");
  AQ_TRY(aq_need_device(device));
  if (len == 0) return AQ_OK;
  double *dx = nullptr, *dx2 = nullptr, *dout = nullptr;
  AQ_HIP(hipMalloc((void **)&dx, len * sizeof(double)));
  AQ_HIP(hipMalloc((void **)&dout, len * sizeof(double)));
  AQ_HIP(hipMemcpy(dx, x, len * sizeof(double), hipMemcpyHostToDevice));
  if (x2) {
    AQ_HIP(hipMalloc((void **)&dx2, len * sizeof(double)));
    AQ_HIP(hipMemcpy(dx2, x2, len * sizeof(double), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(aq_k_special_eval, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, 0, which, dx, dx2, dout, (long long)len);
  hipError_t e = hipMemcpy(out, dout, len * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dout);
  if (dx2) hipFree(dx2);
  if (e != hipSuccess) return aq_fail(AQ_ERR_DEVICE, std::string("aq_special_eval_device: ") + hipGetErrorString(e));
  return AQ_OK;
}

extern "C" int aq_q_approx_vec(const double *x, double *out, int64_t len, int32_t *iters) {
  if (!x || !out || len < 0) return aq_fail(AQ_ERR_ARG, "aq_q_approx_vec: bad argument");
  unsigned long long m0 = ~0ull, m1 = ~0ull;
  bool any = false;
  for (int64_t i = 0; i < len; i++) {
    if (x[i] > 1.0) {
      AqLentz s;
      aq_lentz_init(&s);
      unsigned long long a0 = 0, a1 = 0;
      for (int it = 0; it < 128; it++) {
        double d = aq_lentz_step(&s, x[i], it + 2);
        if (d < 1e-7) { if (it < 64) a0 |= 1ull << it; else a1 |= 1ull << (it - 64); }
      }
      m0 &= a0; m1 &= a1;
      any = true;
    }
  }
  int nit = 0;
  if (any) nit = m0 ? __builtin_ffsll((long long)m0) : (m1 ? 64 + __builtin_ffsll((long long)m1) : 129);
  for (int64_t i = 0; i < len; i++) {
    if (x[i] <= 1.0) {
      out[i] = aq_expint_e1_small(x[i]) * exp(x[i]);
    } else {
      AqLentz s;
      aq_lentz_init(&s);
      for (int it = 0; it < nit; it++) aq_lentz_step(&s, x[i], it + 2);
      out[i] = aq_lentz_finish(&s, x[i]);
    }
  }
  if (iters) *iters = nit;
  return AQ_OK;
}
