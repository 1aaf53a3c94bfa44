// aq_trait_wave.h -- generic core sweep kernel: unblocked n-space Gauss-Seidel, one WAVE per trait.
//
// Used where the blocked f64-MFMA kernels (aq_core_sweep_la.h / aq_core_sweep.h) do not apply:
//   * Y with missing values -- coreDualMisLoop, src/coreLoop.cpp:91-138: the Gram matrix becomes trait
//     specific (cp_X - cp_X_rm[[k]]), which in n-space is just a masked residual:
//         R_ik = mis_ik (y_ik - sum_j x_ij beta_jk),   x_j'R_k + X_norm_sq(j,k) m1 = cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1 (cp_X - cp_X_rm_k)(j,j))
//   * n beyond the register-resident residual tile of the MFMA kernels.
// A workgroup = 16 waves.  WPT = 1: one tile of 16 traits, wave w owns trait w and its whole residual column in
// VGPRs (lane l holds samples l, l+64, ...).  WPT = 2, 4 (n > 2048): WPT workgroups share a tile, each takes 16/WPT
// of its traits and WPT waves split a trait's samples; their partial dots meet in LDS (one barrier per SNP).
// x_j is staged through LDS `ns` SNPs at a time and shared by the 16 waves.
// Per SNP: dot with the (masked) residual -> wave reduction -> the scalar update of src/coreLoop.cpp:121-130
// (every lane redundantly) -> masked AXPY on the residual.  All per-(j,k) inputs/outputs use the same trait-tiled arrays
// as the MFMA kernels, so the rest of the sweep (pre-pass, reductions, p-/q-vector kernels, ELBO) is shared.
#pragma once
#include <hip/hip_runtime.h>
#include "aq_special.h"
#include "aq_vec_kernels.h"

struct AqTwArgs {
  const double *X;       // n x p column-major (standardised)
  double *R;             // [ntile][n_pad][16] residual (masked)
  const double *mis;     // [ntile][n_pad][16] 1 = observed, 0 = missing / padding
  double *XN;            // [ntile][p_pad][16] X_norm_sq(j,k) = sum_i x_ij^2 mis_ik   (R/atlasqtl_global_local_core.R:23)
  double *gam, *mu;      // [ntile][p_pad][16]
  const double *Aarr, *Barr;
  const double *tau, *log_tau, *sig2b;   // [q_pad]; sig2b = initial sig2_beta_vb (init mode only)
  const AqScalars *sc;   // sig2_inv, log_sig2_inv of this sweep
  double *sums;          // [6][q_pad]: sum gam, sum m2, sx (see aq_k_qpre), sum gam*b, ||R||^2, sum gam*log sig2_beta
  double *rowGB;         // [ntile * WPT][p_pad]
  double c;
  int n, p, q, n_pad, p_pad, q_pad, ntile;
  int ns;                // SNP columns staged in LDS at a time (4, 2 or 1; ns * n_pad doubles)
  int mode;              // 0 sweep, 1 init: R = mis .* (Y - X beta) from R = mis .* Y, XN, initial sums
  int complete;          // 1: no missing value anywhere: sig2_beta_vb is the q-vector 1/(c (n-1+sig2_inv) tau) (R/update_vb.R:38)
};

template <int NE, int WPT>   // samples per lane, waves per trait: n_pad = 64 * NE * WPT
__global__ __launch_bounds__(1024) void aq_trait_wave_kernel(const AqTwArgs a) {
  constexpr int TPW = 16 / WPT;                  // traits per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tile = blockIdx.x / WPT, sub = blockIdx.x - tile * WPT;
  const int part = wv % WPT;                     // which slice of the samples this wave holds
  const int w = sub * TPW + wv / WPT;            // trait within the tile
  const bool lead = (part == 0);                 // the wave that records the trait's results
  const int n_pad = 64 * NE * WPT;
  const int ibase = part * 64 * NE + lane;       // this lane's samples: ibase + 64 e
  extern __shared__ double lds[];
  double *xs = lds;                              // [ns][n_pad]   staged SNP columns
  double *LAs = xs + a.ns * n_pad;               // [16][16] A
  double *LBs = LAs + 256, *Lg = LBs + 256, *Lm = Lg + 256, *Lxn = Lm + 256;   // b, gam_old, mu_old, X_norm_sq
  double *Og = Lxn + 256, *Om = Og + 256, *Ogb = Om + 256;                    // outputs: gam, mu, gam*b
  double *Lpart = Ogb + 256;                     // [2][16] partial dots of the 16 waves, double-buffered by SNP parity
  int par = 0;
  // sum of the WPT partial values of this trait, identical (same order) in each of its waves
  auto trait_sum = [&](double v) -> double {
    if (WPT == 1) return v;
    if (lane == 0) Lpart[par * 16 + wv] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int pp = 0; pp < WPT; pp++) s += Lpart[par * 16 + (wv / WPT) * WPT + pp];
    par ^= 1;
    return s;
  };

  const int k = tile * 16 + w;
  const bool kvalid = k < a.q;
  const size_t tbase = (size_t)tile * a.p_pad * 16;
  double *Rg = a.R + (size_t)tile * a.n_pad * 16;
  const double *Mg = a.mis + (size_t)tile * a.n_pad * 16;

  double R[NE];
  unsigned long long mbits = 0;   // NE <= 64 mask bits of this lane's samples
#pragma unroll
  for (int e = 0; e < NE; e++) {
    int i = ibase + 64 * e;
    R[e] = Rg[(size_t)i * 16 + w];
    if (Mg[(size_t)i * 16 + w] != 0.0) mbits |= (1ull << e);
  }
  const double tau = a.tau[k], c = a.c;
  const double sig2_inv = a.sc->sig2_inv;
  const double cstna = -(a.log_tau[k] + a.sc->log_sig2_inv) / 2;            // src/coreLoop.cpp:108
  const double s2_init = a.sig2b[k];
  const double nm1 = (double)(a.n - 1);
  double sum_g = 0, sum_m2 = 0, sum_x = 0, sum_gb = 0, sum_lg = 0;

  const int nb = a.p_pad / 16;
  const int ns = a.ns;
  for (int b = 0; b < nb; b++) {
    // ---- stage the block's per-(j,k) scalars (threads 0..255: entry (j = tid>>4, trait = tid&15)) ----
    __syncthreads();
    if (tid < 256) {
      size_t off = tbase + (size_t)(16 * b) * 16 + tid;
      Lg[tid] = a.gam[off];
      Lm[tid] = a.mu[off];
      Ogb[tid] = 0.0;                           // traits of the tile owned by another workgroup add nothing to rowGB
      if (a.mode == 0) {
        LAs[tid] = a.Aarr[off];
        LBs[tid] = a.Barr[off];
        Lxn[tid] = a.XN[off];
      }
    }
    for (int j4 = 0; j4 < 16; j4 += ns) {
      // ---- stage ns SNP columns ----
      __syncthreads();
      for (int e = tid; e < ns * n_pad; e += 1024) {
        int jj = e / n_pad, i = e - jj * n_pad;
        int j = 16 * b + j4 + jj;
        xs[e] = (j < a.p && i < a.n) ? a.X[(size_t)i + (size_t)a.n * j] : 0.0;
      }
      __syncthreads();
#pragma unroll 1
      for (int jj = 0; jj < ns; jj++) {
        const int jl = j4 + jj;                 // SNP within the block
        const int j = 16 * b + jl;
        const double *xj = xs + jj * n_pad + ibase;
        auto xm = [&](int e) -> double { return ((mbits >> e) & 1ull) ? xj[64 * e] : 0.0; };   // x_ij mis_ik
        if (a.mode == 1) {
          // init: X_norm_sq(j,k), R -= beta_jk x_j (masked), initial column sums
          double gm = Lg[jl * 16 + w], mu = Lm[jl * 16 + w];
          double be = gm * mu;
          double xn = 0.0;
#pragma unroll
          for (int e = 0; e < NE; e++) {
            double xv = xm(e);
            xn += xv * xv;
            R[e] -= be * xv;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) xn += __shfl_xor(xn, o, 64);
          xn = trait_sum(xn);
          if (lane == 0 && lead) {
            Lxn[jl * 16 + w] = xn;
            if (kvalid && j < a.p) {
              double m2 = (mu * mu + s2_init) * gm;          // first m2_beta uses the initial q-vector sig2_beta_vb, :113
              sum_g += gm;
              sum_m2 += m2;
              sum_x += a.complete ? be * be : xn * (m2 - be * be);
            }
          }
          continue;
        }
        // ---- x_j' R_k ----
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int e = 0; e < NE; e += 4) {
          a0 += xm(e) * R[e];
          if (e + 1 < NE) a1 += xm(e + 1) * R[e + 1];
          if (e + 2 < NE) a2 += xm(e + 2) * R[e + 2];
          if (e + 3 < NE) a3 += xm(e + 3) * R[e + 3];
        }
        double s = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        s = trait_sum(s);
        // ---- the update of src/coreLoop.cpp:121-130 (every lane, identical values) ----
        const double m1o = Lg[jl * 16 + w] * Lm[jl * 16 + w];
        const double xn = Lxn[jl * 16 + w];
        const double s2 = a.complete ? 1.0 / (c * (nm1 + sig2_inv) * tau) : 1.0 / (c * (xn + sig2_inv) * tau);   // update_sig2_beta_vb_, R/update_vb.R:38,45
        const double ls2 = log(s2);
        const double mu = c * s2 * tau * (s + m1o * xn);                                        // :125
        const double arg = c * (LAs[jl * 16 + w] - mu * mu / (2 * s2) - ls2 / 2 + cstna);       // :127-129
        const double gm = aq_sigmoid_neg(arg);
        const double m1 = gm * mu;
        const double dl = m1 - m1o;
#pragma unroll
        for (int e = 0; e < NE; e++) R[e] -= dl * xm(e);                                        // :132 in n-space
        if (lane == 0 && lead) {
          double gb = 0.0;
          if (kvalid && j < a.p) {
            double m2 = (mu * mu + s2) * gm;                  // update_m2_beta_, R/update_vb.R:19-31
            gb = gm * LBs[jl * 16 + w];
            sum_g += gm;
            sum_m2 += m2;
            sum_x += a.complete ? m1 * m1 : xn * (m2 - m1 * m1);   // kappa's X_norm_sq terms, R/update_vb.R:152-154
            sum_gb += gb;
            sum_lg += gm * ls2;
          }
          Og[jl * 16 + w] = gm;
          Om[jl * 16 + w] = mu;
          Ogb[jl * 16 + w] = gb;
        }
      }
    }
    // ---- write the block's results (threads 0..255) ----
    __syncthreads();
    if (tid < 256) {
      size_t off = tbase + (size_t)(16 * b) * 16 + tid;
      const bool mine = ((tid & 15) / TPW) == sub;   // this workgroup's traits
      if (a.mode == 1) {
        if (mine) a.XN[off] = Lxn[tid];
      } else {
        if (mine) {
          a.gam[off] = Og[tid];
          a.mu[off] = Om[tid];
        }
        double gb = Ogb[tid];
        gb += __shfl_xor(gb, 8, 64);
        gb += __shfl_xor(gb, 4, 64);
        gb += __shfl_xor(gb, 2, 64);
        gb += __shfl_xor(gb, 1, 64);
        if ((tid & 15) == 0) a.rowGB[(size_t)blockIdx.x * a.p_pad + 16 * b + (tid >> 4)] = gb;
      }
    }
  }
  // ---- residual back, ||R_k||^2, column sums ----
  double rn = 0.0;
#pragma unroll
  for (int e = 0; e < NE; e++) {
    Rg[(size_t)(ibase + 64 * e) * 16 + w] = R[e];
    rn += R[e] * R[e];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rn += __shfl_xor(rn, o, 64);
  __syncthreads();
  rn = trait_sum(rn);
  if (lane == 0 && lead) {
    size_t Q = a.q_pad;
    a.sums[k] = sum_g;
    a.sums[Q + k] = sum_m2;
    a.sums[2 * Q + k] = sum_x;
    a.sums[3 * Q + k] = sum_gb;
    a.sums[4 * Q + k] = rn;
    a.sums[5 * Q + k] = sum_lg;
  }
}
