// aq_core_sweep.h -- argument block, MFMA wrapper and layout notes shared by the blocked f64-MFMA sweep kernels
// (aq_core_sweep_la.h: complete Y, n <= 1056; aq_core_sweep_mis.h: Y with missing values and larger n).
//
// One VB sweep of the spike-and-slab updates (reference src/coreLoop.cpp:38-86, called from
// R/atlasqtl_global_local_core.R:167) fused with the column/row sums of the p x q passes that follow it in the
// reference driver (m2_beta R/update_vb.R:19, Z R/update_vb.R:217-234; S1-S5, S17, S19 need only their sums).
//
// Formulation (n-space, blocked Gauss-Seidel, exactly equivalent in exact arithmetic):
//   for a tile of 16 traits K the workgroup keeps the residual R_K = Y_K - X beta_K
//   (n_pad x 16 fp64) in VGPRs, split by samples over its waves.  For each block b of
//   16 consecutive SNPs:
//     S  = X_b' R_K                       16x16, f64 MFMA 16x16x4, k = samples
//     sequential pass over the 16 SNPs with lane = trait, using the precomputed
//       diagonal Gram block G_b = X_b'X_b:  s_j -= sum_{i<j} G_ji delta_i
//       mu, gam, m1 as src/coreLoop.cpp:69-79
//     R_K -= X_b delta                    f64 MFMA, k = SNPs
//
// Device layouts:
//   XA  [nb][NTT][2][64] double2 : A operand of S  (lane: snp = l&15, sample slot g = l>>4;
//                                   the pair holds k-steps 2h, 2h+1 of the 16-sample tile)
//   XU  [nb][NTT][2][64] double2 : A operand of the update (lane: sample = l&15, snp = 4s + (l>>4);
//                                   the pair holds s = 2h, 2h+1)
//   G   [nb][16][16]             : X_b'X_b
//   R   [ntile][n_pad][16], gam/mu [ntile][p_pad][16]  (trait-tiled, 128 B rows)
#pragma once
#include <hip/hip_runtime.h>
#include "aq_special.h"

typedef double aq_d4 __attribute__((ext_vector_type(4)));

struct AqCoreArgs {
  const double2 *XA;
  const double2 *XU;
  const double *G;
  const double *Gx;      // [nb][16][16]  X_b'X_{b-1} (look-ahead kernel only)
  double *R;          // [ntile][n_pad][16]
  double *gam;        // [ntile][p_pad][16]
  double *mu;         // [ntile][p_pad][16]
  const double *Aarr;    // [ntile][p_pad][16]  log(1-Phi) - log Phi of theta_j + zeta_k   (pre-pass; not read by the look-ahead kernel)
  const double *Barr;    // [ntile][p_pad][16]  slope of Z in gam                          (pre-pass; not read by the look-ahead kernel)
  const double *coef;    // [q_pad]  c*sig2_beta*tau
  const double *inv2s;   // [q_pad]  1/(2 sig2_beta)
  const double *cst;     // [q_pad]  -(log_tau + log_sig2_inv + log sig2_beta)/2
  const double *sig2b;   // [q_pad]
  double *sums;          // [5][q_pad]: sum gam, sum m2, sum beta^2, sum gam*b, ||R||^2
  double *rowGB;         // [ntile][p_pad] partial row sums of gam*b
  double c;
  int p, q;              // true sizes (for masking the padding)
  int p_pad, q_pad, n_pad, nb, ntile;
  int dmode;             // f64 MFMA D layout: 0 -> row = (l>>4) + 4*reg, 1 -> row = 4*(l>>4) + reg
  int mode;              // 0 = full sweep, 1 = init: R -= X (gam*mu) only
  int nseg;              // > 1: chained-segment launch, workgroup s*nwg + k = SNP segment s of trait-tile group k (column sums of
                         // segment s go to sums[s][5][q_pad])
  int stagger;           // look-ahead kernel: matrix waves 4-6 start a phase when their SIMD partner has issued this many tiles (0 = off)
  int *done;             // chained segments: done[group] = number of that group's segments already finished
  int *errflag;          // set when a bounded wait on done[] expires (results invalid, reported to the host)
  // sample split of the look-ahead kernel (n beyond one workgroup's registers): C workgroups share a trait group, workgroup
  // k*C + part holds the residual rows [part, part + 1) * n_pad / C.  Per SNP block each publishes its partial S' and adds the
  // others' in fixed order before running the SAME chain (bitwise identical delta in all parts)
  int C;                 // parts per trait group (1 = no split)
  int xtouch;            // helper waves warm the L2 with the next phase's X operand panels (one 128-B line per lane)
  int xhelper;           // sample split: 1 = the helper wave exchanges the partial S' a block ahead, 0 = the recurrence wave at chain start
  double *Pbuf;          // [nwg][2][C][256 TT] partial S' of each part, double-buffered by block parity
  int *pflag;            // [nwg][C] number of blocks whose partial S' this part has published
  double *rnpart;        // [C][q_pad] partial ||R_k||^2 of each part
  // Y with missing values in the look-ahead kernel (MASK instances): the residual is re-masked after every update, and the
  // Gram blocks of the recursion are the trait's own, X_b' diag(mis_k) X_b and X_b' diag(mis_k) X_{b-1}, precomputed once into
  // HBM (they depend on X and the missingness pattern only): GK[tile][b] = { diag, lower triangle [136][16 traits] ;
  // cross [16][16][16 traits] } = 6272 doubles
  const double *mis;     // [ntile][n_pad][16] 1 = observed
  const double *GK;
  const double *tau, *log_tau;             // [q_pad]
  const double *sig2_inv_p, *log_sig2_inv_p;   // scalars of this sweep (device memory: written by aq_k_qpre just before)
  long long *dbg;        // -DAQ_DIAG_TIME builds only: per (workgroup, wave) cycles spent waiting / in total (tools/prof_roles.sh)
  const double *theta;   // look-ahead kernel (fused pre-pass): theta_vb [p_pad], zeta_vb [q_pad] of this sweep
  const double *zeta;
  double sqrt_c;         // annealing: the Mills ratios are taken at sqrt(c) (theta_j + zeta_k), R/update_vb.R:219-224
  int c_is_one;
  int mprio;             // look-ahead kernel: matrix waves run their hand-offs (everything outside the MFMA stream) at raised priority
  int hprio;             // look-ahead kernel: s_setprio level of the helper wave (its fp64 VALU work shares SIMD 3 with the recurrence wave's MFMAs)
};

// Look-ahead kernel: 16-sample residual tiles owned by the RECURRENCE wave (on top of the 3 (NT + NT2) of the six matrix
// waves).  With two trait tiles per workgroup a phase is long enough for that wave to run its chain and then some matrix
// work on SIMD 3, which otherwise issues no MFMA at all.  Shared by the kernel template and the host's geometry.
constexpr int AQ_GK_DIAG = 136 * 16, AQ_GK_STRIDE = 136 * 16 + 256 * 16;   // doubles per (tile, SNP block) of AqCoreArgs::GK
constexpr int aq_la_nt3(int NT, int NT2, int TT) { return (TT == 2 && NT >= 8) ? (NT2 == NT ? 3 : 6) : 0; }

__device__ __forceinline__ aq_d4 aq_mfma(double a, double b, aq_d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// Workgroup barrier that waits for this wave's LDS traffic only: global loads issued before it
// (operand prefetch, staging) stay in flight across it.  __syncthreads() would drain vmcnt too.
__device__ __forceinline__ void aq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int aq_drow(int dmode, int reg, int g) { return (dmode ? 1 : 4) * reg + (dmode ? 4 : 1) * g; }

