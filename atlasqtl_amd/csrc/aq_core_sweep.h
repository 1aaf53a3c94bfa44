// aq_core_sweep.h -- the hot kernel: one VB sweep of the spike-and-slab updates
// (reference src/coreLoop.cpp:38-86, called from R/atlasqtl_global_local_core.R:167)
// fused with the column/row sums of the p x q passes that follow it in the reference
// driver (m2_beta R/update_vb.R:19, Z R/update_vb.R:217-234; S1-S5, S17, S19 need only
// their sums).  The transcendental part of those passes lives in aq_k_prepass
// (aq_vec_kernels.h): Z is linear in gam, Z = a + gam*b, and the pre-pass stores
// A = log(1-Phi) - log Phi and the slope b per entry.
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
//   The update of block b and the S of block b+1 are issued tile by tile while the
//   16-sample residual tile sits in the MFMA accumulator registers.
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

#ifndef AQ_DIAG
#define AQ_DIAG 0      // timing diagnostics only (wrong results): 1 = skip the sequential pass, 2 = skip the MFMAs, 4 = no probit math in the helper wave, 8 = record per-phase time stamps of workgroup 0 into AqCoreArgs::dbg
#endif

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
  int tile_first;        // look-ahead kernel: first trait tile of this launch
  int b_begin, b_end;    // look-ahead kernel: SNP blocks [b_begin, b_end) handled by this launch (one segment)
  int sums_slot;         // per-segment slot of the column sums: sums[slot][5][q_pad]
  int nseg;              // > 1: chained-segment launch, block s*ntile + k = SNP segment s of trait tile k
  int *done;             // chained segments: done[tile] = number of that tile's segments already finished
  int *errflag;          // set when a bounded wait on done[] expires (results invalid, reported to the host)
  long long *dbg;        // AQ_DIAG & 8 only: per-phase s_memtime stamps of workgroup 0 (tools/dev_check.py phases)
  const double *theta;   // look-ahead kernel (fused pre-pass): theta_vb [p_pad], zeta_vb [q_pad] of this sweep
  const double *zeta;
  double sqrt_c;         // annealing: the Mills ratios are taken at sqrt(c) (theta_j + zeta_k), R/update_vb.R:219-224
  int c_is_one;
};

__device__ __forceinline__ aq_d4 aq_mfma(double a, double b, aq_d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// Workgroup barrier that waits for this wave's LDS traffic only: global loads issued before it
// (operand prefetch, staging) stay in flight across it.  __syncthreads() would drain vmcnt too.
__device__ __forceinline__ void aq_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int aq_drow(int dmode, int reg, int g) { return (dmode ? 1 : 4) * reg + (dmode ? 4 : 1) * g; }

// NT: 16-sample residual tiles per wave; NW: waves per workgroup; TT: 16-trait tiles per workgroup.
template <int NT, int NW, int TT>
__global__ __launch_bounds__(NW * 64, (TT == 1 ? 2 : 1)) void aq_core_sweep_kernel(const AqCoreArgs a) {
  constexpr int NTT = NT * NW;
  constexpr int NTR = 16 * TT;     // traits per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int g = lane >> 4;     // k slot of the MFMA operands / row group of D
  const int col = lane & 15;   // trait within a tile (B / D column), row of A
  const int tile0 = blockIdx.x * TT;
  const bool helper = tid < 256;
  const int hj = (tid >> 4) & 15, hk = tid & 15;

  __shared__ double Sp[NW][TT][256];   // per-wave partial S  [tile][snp][trait]
  __shared__ double LA[TT][256];       // A of the block  [snp][trait]
  __shared__ double Lm1[TT][256];      // old m1 = gam*mu
  __shared__ double LG[512];           // Gram block [16][32]: row j holds G[j][0..15] then 16 zeros
  __shared__ double Lgam[TT][256], Lmu[TT][256], Ldel[TT][256];
  __shared__ double Lred[4][TT][256];   // per-thread running column sums (gam, m2, beta^2, gam*b)
  __shared__ double LB[TT][256];        // slope b of Z for the block
  __shared__ double Lrn[NW * 4][TT][16];

  // ---- residual tiles into registers: Rr[tt][t][r] <-> sample 16*(w*NT+t) + drow(r,g), trait col
  aq_d4 Rr[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; tt++) {
    const bool tv = (tile0 + tt) < a.ntile;
    const double *Rg = a.R + (size_t)(tv ? tile0 + tt : 0) * a.n_pad * 16;
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int s = 16 * (w * NT + t) + aq_drow(a.dmode, r, g);
        Rr[tt][t][r] = tv ? Rg[(size_t)s * 16 + col] : 0.0;
      }
    }
  }

  const double2 *XAw = a.XA + (size_t)(w * NT) * 128 + lane;   // + (b*NTT + t)*128 + h*64
  const double2 *XUw = a.XU + (size_t)(w * NT) * 128 + lane;

  // helper-thread state: entry (hj, hk) of each of the TT tiles
  double sig2b_k[TT];
  bool kvalid[TT], tvalid[TT];
  size_t tbase[TT];
#pragma unroll
  for (int tt = 0; tt < TT; tt++) {
    if (helper) Lred[0][tt][tid] = Lred[1][tt][tid] = Lred[2][tt][tid] = Lred[3][tt][tid] = 0.0;
    tvalid[tt] = (tile0 + tt) < a.ntile;
    int kk = (tile0 + tt) * 16 + hk;
    kvalid[tt] = tvalid[tt] && kk < a.q;
    sig2b_k[tt] = (helper && tvalid[tt]) ? a.sig2b[kk] : 1.0;
    tbase[tt] = (size_t)(tvalid[tt] ? tile0 + tt : 0) * a.p_pad * 16;
  }
  if (helper) LG[hj * 32 + 16 + hk] = 0.0;

  if (a.mode == 1) {
    // ---------------- init mode: R -= X (gam*mu), block by block -----------------
    for (int b = 0; b < a.nb; b++) {
      if (helper) {
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
          size_t off = tbase[tt] + (size_t)(16 * b) * 16 + tid;
          double gm = tvalid[tt] ? a.gam[off] : 0.0, mu = tvalid[tt] ? a.mu[off] : 0.0;
          double be = gm * mu;                                // update_beta_vb_, R/update_vb.R:17
          Ldel[tt][tid] = be;
          if (kvalid[tt] && (16 * b + hj) < a.p) {
            Lred[0][tt][tid] += gm;
            Lred[1][tt][tid] += (mu * mu + sig2b_k[tt]) * gm;   // update_m2_beta_ with the initial sig2_beta_vb, :113
            Lred[2][tt][tid] += be * be;
          }
        }
      }
      __syncthreads();
      double nd[TT][4];
#pragma unroll
      for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int s = 0; s < 4; s++) nd[tt][s] = -Ldel[tt][(4 * s + g) * 16 + col];
      const double2 *xu = XUw + (size_t)b * NTT * 128;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        double2 u0 = xu[t * 128], u1 = xu[t * 128 + 64];
#pragma unroll
        for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(u0.x, nd[tt][0], Rr[tt][t]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(u0.y, nd[tt][1], Rr[tt][t]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(u1.x, nd[tt][2], Rr[tt][t]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(u1.y, nd[tt][3], Rr[tt][t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else {
    // ---------------- full sweep ------------------------------------------------
    // helper threads stage everything of block b that does not depend on the recursion:
    // global loads are issued a whole MFMA phase ahead (stage_load) and written to LDS
    // after it (stage_commit), so their latency never sits on the block's critical path.
    double st_A[TT], st_g[TT], st_m[TT], st_B[TT], st_G = 0.0;
#pragma unroll
    for (int tt = 0; tt < TT; tt++) st_A[tt] = st_g[tt] = st_m[tt] = st_B[tt] = 0.0;
    auto stage_load = [&](int b) {
#pragma unroll
      for (int tt = 0; tt < TT; tt++) {
        size_t off = tbase[tt] + (size_t)(16 * b) * 16 + tid;
        st_A[tt] = a.Aarr[off];
        st_g[tt] = a.gam[off];
        st_m[tt] = a.mu[off];
        st_B[tt] = a.Barr[off];
      }
      st_G = a.G[(size_t)b * 256 + tid];
    };
    auto stage_commit = [&]() {
#pragma unroll
      for (int tt = 0; tt < TT; tt++) {
        LA[tt][tid] = st_A[tt];
        Lm1[tt][tid] = st_g[tt] * st_m[tt];
        LB[tt][tid] = st_B[tt];
      }
      LG[hj * 32 + hk] = st_G;
    };
    aq_d4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; tt++) acc[tt] = (aq_d4){0, 0, 0, 0};
    {
      const double2 *xa = XAw;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        double2 a0 = xa[t * 128], a1 = xa[t * 128 + 64];
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(a0.x, Rr[tt][t][0], acc[tt]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(a0.y, Rr[tt][t][1], acc[tt]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(a1.x, Rr[tt][t][2], acc[tt]);
#pragma unroll
        for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(a1.y, Rr[tt][t][3], acc[tt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (helper) { stage_load(0); stage_commit(); }

    for (int b = 0; b < a.nb; b++) {
      const bool more = (b + 1 < a.nb);
      // the staging loads of block b+1 stay in flight across the sequential pass
      if (helper && more) stage_load(b + 1);
      // partial S of this wave -> LDS
#pragma unroll
      for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int i = 0; i < 4; i++) Sp[w][tt][aq_drow(a.dmode, i, g) * 16 + col] = acc[tt][i];
      aq_lds_barrier();

      // ---- sequential pass over the 16 SNPs of the block, lane = trait (wave 0) ----
      if (w == 0 && lane < NTR && !(AQ_DIAG & 1)) {
        const int rt = lane >> 4;                          // tile of this lane's trait
        const int kk = (tile0 + rt < a.ntile ? tile0 + rt : 0) * 16 + col;
        const double rc_coef = a.coef[kk];
        const double rc_cinv2s = a.c * a.inv2s[kk];
        const double rc_cst = a.cst[kk];
        const double rc_K = rc_coef * rc_coef * rc_cinv2s;
        // S[0] is always the SNP being visited: after each step the vector shifts down by one
        // while the in-block Gram correction is applied (rows past the block read the zero pad).
        double S[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          double s = Sp[0][rt][j * 16 + col];
#pragma unroll
          for (int ww = 1; ww < NW; ww++) s += Sp[ww][rt][j * 16 + col];
          S[j] = s;
        }
        double m1o = Lm1[rt][col], cA = a.c * (LA[rt][col] + rc_cst), dj = LG[0];
#pragma unroll 1
        for (int j = 0; j < 16; j++) {
          const int jn = (j + 1) & 15;
          double m1o_n = Lm1[rt][jn * 16 + col], cA_n = a.c * (LA[rt][jn * 16 + col] + rc_cst), d_n = LG[jn * 33];
          double s = S[0] + m1o * dj;                       // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1*cp_X(j,j))   :71
          double mu = rc_coef * s;                          // :73
          double x = fma(-(s * s), rc_K, cA);               // c*(log(1-Phi) - log Phi - mu^2/(2 sig2) + cst), mu^2 = coef^2 s^2   :75-77
          double gm = aq_sigmoid_neg_fast(x);
          double dl = gm * mu - m1o;                        // m1 - m1_old, m1 = gam*mu   :79
          // in-block part of :81: S shifts down by one; G[j+1+i][j] (symmetric) reads the zero pad past the block
#pragma unroll
          for (int i0 = 0; i0 < 15; i0 += 8) {
            double gc[8];
#pragma unroll
            for (int i = 0; i < 8; i++) gc[i] = (i0 + i < 15) ? LG[j * 32 + j + 1 + i0 + i] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++)
              if (i0 + i < 15) S[i0 + i] = S[i0 + i + 1] - gc[i] * dl;
            __builtin_amdgcn_sched_barrier(0);
          }
          Lgam[rt][j * 16 + col] = gm;
          Lmu[rt][j * 16 + col] = mu;
          Ldel[rt][j * 16 + col] = dl;
          m1o = m1o_n; cA = cA_n; dj = d_n;
        }
      }
      aq_lds_barrier();

      // ---- finalize block b (helper threads): stores and column/row sums ----
      if (helper) {
        const int j = 16 * b + hj;
#pragma unroll
        for (int tt = 0; tt < TT; tt++) {
          double gm = Lgam[tt][tid], mu = Lmu[tt][tid];
          size_t off = tbase[tt] + (size_t)(16 * b) * 16 + tid;
          double gb = 0.0;
          if (tvalid[tt]) {
            a.gam[off] = gm;
            a.mu[off] = mu;
            if (kvalid[tt] && j < a.p) {
              double be = gm * mu;
              gb = gm * LB[tt][tid];
              Lred[0][tt][tid] += gm;
              Lred[1][tt][tid] += (mu * mu + sig2b_k[tt]) * gm;   // update_m2_beta_, R/update_vb.R:19-31
              Lred[2][tt][tid] += be * be;
              Lred[3][tt][tid] += gb;
            }
          }
          // row sum over the 16 traits of this tile (16-lane groups are aligned)
          gb += __shfl_xor(gb, 8, 64);
          gb += __shfl_xor(gb, 4, 64);
          gb += __shfl_xor(gb, 2, 64);
          gb += __shfl_xor(gb, 1, 64);
          if (hk == 0 && tvalid[tt]) a.rowGB[(size_t)(tile0 + tt) * a.p_pad + j] = gb;
        }
      }

      // ---- R -= X_b delta, and S of block b+1, tile by tile (S one tile behind) ----
      double nd[TT][4];
#pragma unroll
      for (int tt = 0; tt < TT; tt++)
#pragma unroll
        for (int s = 0; s < 4; s++) nd[tt][s] = -Ldel[tt][(4 * s + g) * 16 + col];
#pragma unroll
      for (int tt = 0; tt < TT; tt++) acc[tt] = (aq_d4){0, 0, 0, 0};
      constexpr bool do_mfma = !(AQ_DIAG & 2);
      const double2 *xu = XUw + (size_t)b * NTT * 128;
      const double2 *xa = XAw + (size_t)(more ? b + 1 : b) * NTT * 128;
      double2 cu0 = xu[0], cu1 = xu[64], ca0 = xa[0], ca1 = xa[64];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        double2 nu0, nu1, na0, na1;
        if (t + 1 < NT) {
          nu0 = xu[(t + 1) * 128]; nu1 = xu[(t + 1) * 128 + 64];
          na0 = xa[(t + 1) * 128]; na1 = xa[(t + 1) * 128 + 64];
        }
        if (do_mfma) {
#pragma unroll
          for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(cu0.x, nd[tt][0], Rr[tt][t]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(cu0.y, nd[tt][1], Rr[tt][t]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(cu1.x, nd[tt][2], Rr[tt][t]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) Rr[tt][t] = aq_mfma(cu1.y, nd[tt][3], Rr[tt][t]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(ca0.x, Rr[tt][t][0], acc[tt]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(ca0.y, Rr[tt][t][1], acc[tt]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(ca1.x, Rr[tt][t][2], acc[tt]);
#pragma unroll
          for (int tt = 0; tt < TT; tt++) acc[tt] = aq_mfma(ca1.y, Rr[tt][t][3], acc[tt]);
        }
        if (t + 1 < NT) { cu0 = nu0; cu1 = nu1; ca0 = na0; ca1 = na1; }
        __builtin_amdgcn_sched_barrier(0);   // keep later tiles' operand loads from being hoisted (VGPR budget)
      }
      // ---- block b+1's staged values -> LDS (helper threads) ----
      if (helper && more) stage_commit();
    }
  }

  // ---- write the residual back, ||R_k||^2 and the per-trait sums ----
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; tt++) {
    const bool tv = (tile0 + tt) < a.ntile;
    double *Rg = a.R + (size_t)(tv ? tile0 + tt : 0) * a.n_pad * 16;
    double rn = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int s = 16 * (w * NT + t) + aq_drow(a.dmode, r, g);
        double v = Rr[tt][t][r];
        if (tv) Rg[(size_t)s * 16 + col] = v;
        rn += v * v;
      }
    }
    Lrn[w * 4 + g][tt][col] = rn;
  }
  __syncthreads();
  if (tid < NTR) {
    const int tt = tid >> 4, k15 = tid & 15;
    if (tile0 + tt < a.ntile) {
      int kk = (tile0 + tt) * 16 + k15;
      double r2 = 0.0;
      for (int s = 0; s < NW * 4; s++) r2 += Lrn[s][tt][k15];
      a.sums[(size_t)4 * a.q_pad + kk] = r2;
      for (int v = 0; v < 4; v++) {
        double acc2 = 0.0;
        for (int jj = 0; jj < 16; jj++) acc2 += Lred[v][tt][jj * 16 + k15];
        a.sums[(size_t)v * a.q_pad + kk] = acc2;
      }
    }
  }
}
