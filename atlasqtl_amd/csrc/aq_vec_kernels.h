// aq_vec_kernels.h -- layout conversion, p- and q-vector updates, reductions and
// the ELBO pieces that surround the core sweep kernel.  Step numbers (S1..S22)
// are those of SURVEY.md section 3.2; each kernel cites the reference lines it follows.
#pragma once
#include <hip/hip_runtime.h>
#include "aq_special.h"

#define AQ_RED_EXTRA 8   // scalars appended to the row-sum all-reduce payload

// Device-resident scalars of the VB state (one struct, updated by 1-thread kernels).
struct AqScalars {
  double sig02_inv;     // horseshoe global precision sig02_inv_vb
  double S_gam;         // sum(gam_vb) over all traits (all-reduced)
  double T2;            // sum_k tau_k * colSums(m2_beta)_k (all-reduced)
  double sum_zeta_old;  // sum(zeta_vb) before this sweep's zeta update (all-reduced)
  double nu_vb, rho_vb, sig2_inv, log_sig2_inv;   // S1-S3, S8
  double rho_xi_inv, xi_inv, nu_s0, rho_s0;       // S13, S15, S18
  double sum_theta;     // sum(theta_vb) after S17
  double sum_sig2_theta;
  double sum_theta_sq;  // global-only core: sum (theta - m0)^2
  double elbo_C;        // e_theta_hs_ / e_theta_ (replicated p-sum)
  double elbo;          // assembled ELBO
  unsigned long long lentz_mask[2];
  int lentz_iters;
  int pad_;
};

// ---------------------------------------------------------------- layouts ----
__device__ __forceinline__ int aqv_drow(int dmode, int reg, int g) { return dmode ? (4 * g + reg) : (4 * reg + g); }

// X (n x p, R column-major) -> XA / XU MFMA operand layouts (see aq_core_sweep.h)
__global__ void aq_k_build_x_layouts(const double *__restrict__ X, double2 *__restrict__ XA,
                                     double2 *__restrict__ XU, int n, int p, int nb, int NTT, int dmode) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nb * NTT * 128;
  if (e >= total) return;
  int lane = (int)(e & 63);
  int h = (int)((e >> 6) & 1);
  size_t bt = e >> 7;
  int T = (int)(bt % NTT);
  int b = (int)(bt / NTT);
  int g = lane >> 4, c15 = lane & 15;
  {
    int snp = 16 * b + c15;
    int s0 = 16 * T + aqv_drow(dmode, 2 * h, g), s1 = 16 * T + aqv_drow(dmode, 2 * h + 1, g);
    double2 v;
    v.x = (snp < p && s0 < n) ? X[(size_t)s0 + (size_t)n * snp] : 0.0;
    v.y = (snp < p && s1 < n) ? X[(size_t)s1 + (size_t)n * snp] : 0.0;
    XA[e] = v;
  }
  {
    int s = 16 * T + c15;
    int j0 = 16 * b + 4 * (2 * h) + g, j1 = 16 * b + 4 * (2 * h + 1) + g;
    double2 v;
    v.x = (s < n && j0 < p) ? X[(size_t)s + (size_t)n * j0] : 0.0;
    v.y = (s < n && j1 < p) ? X[(size_t)s + (size_t)n * j1] : 0.0;
    XU[e] = v;
  }
}

// diagonal Gram blocks G[b] = X_b' X_b and first off-diagonal blocks Gx[b] = X_b' X_{b-1} (16 x 16 each):
// the only parts of cp_X (R/atlasqtl_global_local_core.R:41) the blocked recursion needs
__global__ void aq_k_gram_blocks(const double *__restrict__ X, double *__restrict__ G, double *__restrict__ Gx, int n,
                                 int p) {
  int b = blockIdx.x;
  int i = threadIdx.x >> 4, j = threadIdx.x & 15;
  int ji = 16 * b + i, jj = 16 * b + j;
  {   // cross block with the previous SNP block: Gx[b][i][j] = x_{16b+i}' x_{16(b-1)+j}
    int jp = 16 * (b - 1) + j;
    double sx = 0.0;
    if (b > 0 && ji < p && jp < p) {
      const double *xi = X + (size_t)n * ji, *xj = X + (size_t)n * jp;
      for (int r = 0; r < n; r++) sx += xi[r] * xj[r];
    }
    Gx[(size_t)b * 256 + threadIdx.x] = sx;
  }
  double s = 0.0;
  if (ji < p && jj < p) {
    const double *xi = X + (size_t)n * ji, *xj = X + (size_t)n * jj;
    if (i <= j) {
      for (int r = 0; r < n; r++) s += xi[r] * xj[r];
    } else {
      for (int r = 0; r < n; r++) s += xj[r] * xi[r];
    }
  }
  G[(size_t)b * 256 + threadIdx.x] = s;
}

// (rows x q) R column-major  ->  [ntile][rows_pad][16] trait-tiled; zero padded.
// grid (ceil(rows_pad/64), ntile), 256 threads.  nan_to_zero: Y with NA -> 0 (R/atlasqtl_global_local_core.R:22)
__global__ void aq_k_tile_from_colmajor(const double *__restrict__ src, double *__restrict__ dst, int rows, int q,
                                        int rows_pad, int nan_to_zero) {
  __shared__ double buf[16][65];
  int tile = blockIdx.y;
  int r0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    int k = e >> 6, rr = e & 63;
    int kk = tile * 16 + k, r = r0 + rr;
    double v = 0.0;
    if (kk < q && r < rows) {
      v = src[(size_t)r + (size_t)rows * kk];
      if (nan_to_zero && v != v) v = 0.0;
    }
    buf[k][rr] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    int rr = e >> 4, k = e & 15;
    int r = r0 + rr;
    if (r < rows_pad) dst[((size_t)tile * rows_pad + r) * 16 + k] = buf[k][rr];
  }
}

// trait-tiled -> column-major (rows x q); mul != NULL gives src*mul (beta_vb = gam_vb * mu_beta_vb, R/update_vb.R:17)
__global__ void aq_k_colmajor_from_tile(const double *__restrict__ src, const double *__restrict__ mul,
                                        double *__restrict__ dst, int rows, int q, int rows_pad) {
  __shared__ double buf[16][65];
  int tile = blockIdx.y;
  int r0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    int rr = e >> 4, k = e & 15;
    int r = r0 + rr;
    double v = 0.0;
    if (r < rows_pad) {
      size_t off = ((size_t)tile * rows_pad + r) * 16 + k;
      v = src[off];
      if (mul) v *= mul[off];
    }
    buf[k][rr] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    int k = e >> 6, rr = e & 63;
    int kk = tile * 16 + k, r = r0 + rr;
    if (kk < q && r < rows) dst[(size_t)r + (size_t)rows * kk] = buf[k][rr];
  }
}

// ----------------------------------------------- initial values on the device ----
// auto_set_init_ (R/set_hyper_init.R:385-387): gam_vb = pnorm(rnorm(p q, mean = n0, sd = s02 + t02)), mu_beta_vb = rnorm(p q).
// R's Mersenne-Twister stream cannot be reproduced anyway, so the draws come from a counter-based generator
// that any shard can evaluate on its own: Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = (SNP j, global
// trait k, 0, 0); one call yields both normals of the entry (Box-Muller).  the CPU checker under tests/ restates the same stream.
__host__ __device__ inline void aq_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__host__ __device__ inline void aq_init_pair(uint64_t seed, uint32_t j, uint32_t k_global, double gam_mean, double gam_sd,
                                             double *gam, double *mu) {
  uint32_t c[4] = {j, k_global, 0u, 0u};
  aq_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  // two uniforms in (0, 1) with 53 bits each
  const double u1 = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6) + 0.5) * (1.0 / 9007199254740992.0);
  const double r = sqrt(-2.0 * log(u1));
  const double z1 = r * cos(6.283185307179586476925286766559 * u2), z2 = r * sin(6.283185307179586476925286766559 * u2);
  *gam = 0.5 * erfc(-(gam_mean + gam_sd * z1) * 0.70710678118654752440084436210485);    // pnorm
  *mu = z2;
}
// gam, mu in the trait-tiled layout [ntile][p_pad][16]; padding entries are 0
__global__ void aq_k_init_generate(double *__restrict__ gam, double *__restrict__ mu, int p, int q, int p_pad, int ntile,
                                   unsigned long long seed, int trait_offset, double gam_mean, double gam_sd) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)ntile * p_pad * 16;
  if (e >= total) return;
  int hk = (int)(e & 15);
  size_t rest = e >> 4;
  int j = (int)(rest % p_pad), tile = (int)(rest / p_pad);
  int k = tile * 16 + hk;
  double g = 0.0, m = 0.0;
  if (j < p && k < q) aq_init_pair(seed, (uint32_t)j, (uint32_t)(trait_offset + k), gam_mean, gam_sd, &g, &m);
  gam[e] = g;
  mu[e] = m;
}

// ------------------------------------------------------------- pre-pass ----
// Everything transcendental that the core sweep needs per (j,k) entry, from the current
// theta_j + zeta_k (u):
//   A = log(1-Phi(u)) - log Phi(u)                     src/coreLoop.cpp:75-76 (its log_Phi / log_1_min_Phi inputs,
//                                                      R/atlasqtl_global_local_core.R:61-63,293-295)
//   Z = a + gam*b with U = sqrt(c) u                   R/update_vb.R:217-234, R/utils.R:172-191
//       a = u + imr0/sqrt(c),  b = (imr1 - imr0)/sqrt(c)
// A and b are stored; the row / column sums of a are reduced here.  With do_H the same
// pass adds the entropy-like p x q part of e_beta_gamma_ (R/elbo.R:10-34) for the ELBO that
// closes the previous sweep (it needs exactly these refreshed log Phi values).
// grid (nchunk, ntile), 256 threads: thread (hj, hk) walks rows hj, hj+16, ... of its chunk.
struct AqPrepass {
  const double *theta, *zeta, *gam;
  double *Aarr, *Barr, *rowA, *colApart, *Hpart;
  int p, q, p_pad, q_pad, rows_per_chunk;
  double sqrt_c;
  int c_is_one, do_H;
  int write_AB;   // 0: only the ELBO part (do_H); A, b and the sums of a are produced inside the sweep kernel
};

__global__ __launch_bounds__(256) void aq_k_prepass(AqPrepass v) {
  __shared__ double sh[256];
  const double eps = 1.81898940354585648e-12;   // .Machine$double.eps^0.75, R/elbo.R:15
  int tile = blockIdx.y, chunk = blockIdx.x;
  int hj = threadIdx.x >> 4, hk = threadIdx.x & 15;
  int kk = tile * 16 + hk;
  bool kvalid = kk < v.q;
  double zk = v.zeta[kk];
  double colA = 0.0, hacc = 0.0;
  int j0 = chunk * v.rows_per_chunk;
  int j1 = min(j0 + v.rows_per_chunk, v.p_pad);
  if (!v.write_AB) {
    // The ELBO's p x q part alone (the sweep kernel produces A, b and the sums of a itself): sum over the entries of
    //   gam log Phi(u) + (1 - gam) log(1 - Phi(u)) - gam log(gam + eps) - (1 - gam) log(1 - gam + eps)        R/elbo.R:10-34
    // log Phi and log(1 - Phi) from the tables (two Horner chains instead of erfcx + exp + log + log1p: 6.5 -> 2.x ms at C3), the
    // entropy terms with the short logarithm.  The tables sit in LDS, coefficient-major: lanes in different intervals read
    // different banks.
    __shared__ double tA[AQ_PT_N_LEN], tN[AQ_PT_N_LEN];
    for (int e = threadIdx.x; e < AQ_PT_N_LEN; e += 256) { tA[e] = aq_pt_dev[e]; tN[e] = aq_ptn_dev[e]; }
    __syncthreads();
    if (v.do_H && kvalid) {
      for (int j = j0 + hj; j < j1 && j < v.p; j += 16) {
        const size_t off = ((size_t)tile * v.p_pad + j) * 16 + hk;
        const double g = v.gam[off];
        double lP, l1;
        aq_log_ndtr_pair_tab(v.theta[j] + zk, tA, tN, &lP, &l1);
        hacc += g * lP + (1 - g) * l1 - g * aq_log_pos(g + eps) - (1 - g) * aq_log_pos(1 - g + eps);
      }
    }
    if (v.do_H) {
      sh[threadIdx.x] = hacc;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
      }
      if (threadIdx.x == 0) v.Hpart[(size_t)tile * gridDim.x + chunk] = sh[0];
    }
    return;
  }
  for (int j = j0 + hj; j < j1; j += 16) {
    size_t off = ((size_t)tile * v.p_pad + j) * 16 + hk;
    double A = 0.0, B = 0.0, aa = 0.0;
    if (j < v.p && kvalid) {
      double u = v.theta[j] + zk;
      double imr1, imr0, e;
      aq_probit_A_imr(u, &A, &imr1, &imr0, &e);
      if (!v.c_is_one) {   // annealed: the Mills ratios are taken at sqrt(c) u, R/update_vb.R:223-224
        double Ac, ec;
        aq_probit_A_imr(v.sqrt_c * u, &Ac, &imr1, &imr0, &ec);
        imr1 /= v.sqrt_c;
        imr0 /= v.sqrt_c;
      }
      aa = u + imr0;
      B = imr1 - imr0;
      colA += aa;
      if (v.do_H) {
        double g = v.gam[off];
        double near_ = log1p(-e);                   // log of the near-side probability; A = l1 - lP
        double lP = u > 0.0 ? near_ : near_ - A;    // u <= 0: lP is the far tail = near - (near - far)
        double l1 = u > 0.0 ? near_ + A : near_;
        hacc += g * lP + (1 - g) * l1 - g * log(g + eps) - (1 - g) * log(1 - g + eps);
      }
    }
    if (v.write_AB) {
      v.Aarr[off] = A;
      v.Barr[off] = B;
      double r = aa;
      r += __shfl_xor(r, 8, 64);
      r += __shfl_xor(r, 4, 64);
      r += __shfl_xor(r, 2, 64);
      r += __shfl_xor(r, 1, 64);
      if (hk == 0) v.rowA[(size_t)tile * v.p_pad + j] = r;
    }
  }
  // column sums of a over the 16 row slots -> colApart[chunk][kk]
  sh[threadIdx.x] = colA;
  __syncthreads();
  if (threadIdx.x < 16 && v.write_AB) {
    double s = 0.0;
    for (int r = 0; r < 16; r++) s += sh[r * 16 + threadIdx.x];
    v.colApart[(size_t)chunk * v.q_pad + tile * 16 + threadIdx.x] = s;
  }
  __syncthreads();
  if (v.do_H) {
    sh[threadIdx.x] = hacc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) v.Hpart[(size_t)tile * gridDim.x + chunk] = sh[0];
  }
}

// -------------------------------------------------------- q-vector: S1-S8 ----
struct AqQvec {
  const double *eta_h, *kappa_h, *n0, *nobs;  // hyper (q_pad), observed-sample counts
  double *zeta, *tau, *sig2b, *log_tau, *eta_vb, *kappa_vb, *coef, *inv2s, *cst;
  double *sums;  // [5][q_pad]: sum gam, sum m2, sum beta^2, sum gam*b, ||R||^2
  const double *colApart;  // [nchunk][q_pad] column sums of the Z intercept a
  int nchunk;
  int q, q_pad, n;
  double nu_h, rho_h;
  int na;   // 1: Y has missing values: kappa uses the X_norm_sq form (R/update_vb.R:150-155), sums[2] = sum_j X_norm_sq (m2 - beta^2),
            //    sums[5] = sum_j gam log sig2_beta_jk (sig2_beta_vb is p x q, R/update_vb.R:45)
};

// S1-S8: R/atlasqtl_global_local_core.R:134-150 with R/update_vb.R:116-159,33-50.
// kappa in n-space: Y_norm_sq - 2 sum beta (Y'X) + sum (X'X beta) beta == ||y_k - X beta_k||^2 = sums[4].
__global__ void aq_k_qpre(AqQvec v, AqScalars *sc, double c) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  double nu_vb = c * (v.nu_h + sc->S_gam / 2) - c + 1;        // update_nu_vb_
  double rho_vb = c * (v.rho_h + sc->T2 / 2);                  // update_rho_vb_
  double sig2_inv = nu_vb / rho_vb;                            // :137
  double log_sig2_inv = aq_digamma(nu_vb) - log(rho_vb);       // update_log_sig2_inv_vb_
  if (k == 0) {
    sc->nu_vb = nu_vb;
    sc->rho_vb = rho_vb;
    sc->sig2_inv = sig2_inv;
    sc->log_sig2_inv = log_sig2_inv;
  }
  if (k >= v.q_pad) return;
  if (k >= v.q) {   // padded trait: benign constants, results never read back
    v.tau[k] = 1.0; v.sig2b[k] = 1.0; v.log_tau[k] = 0.0; v.coef[k] = 0.0; v.inv2s[k] = 0.5; v.cst[k] = 0.0;
    return;
  }
  const double *S = v.sums;
  size_t Q = v.q_pad;
  double sg = S[k], sm2 = S[Q + k], sb2 = S[2 * Q + k], rn = S[4 * Q + k];
  double nm1 = (double)(v.n - 1);
  double eta_vb = c * (v.eta_h[k] + v.nobs[k] / 2 + sg / 2) - c + 1;                       // update_eta_vb_
  double kappa_vb = v.na ? c * (v.kappa_h[k] + (rn + sig2_inv * sm2 + sb2) / 2)
                         : c * (v.kappa_h[k] + (rn + (nm1 + sig2_inv) * sm2 - nm1 * sb2) / 2);    // update_kappa_vb_
  double tau = eta_vb / kappa_vb;                                                          // :145
  double s2b = 1.0 / (c * (nm1 + sig2_inv) * tau);                                         // update_sig2_beta_vb_
  double log_tau = aq_digamma(eta_vb) - log(kappa_vb);                                     // update_log_tau_vb_
  v.eta_vb[k] = eta_vb;
  v.kappa_vb[k] = kappa_vb;
  v.tau[k] = tau;
  v.sig2b[k] = s2b;
  v.log_tau[k] = log_tau;
  v.coef[k] = c * s2b * tau;                                   // src/coreLoop.cpp:73
  v.inv2s[k] = 1.0 / (2 * s2b);
  v.cst[k] = -(log_tau + log_sig2_inv + log(s2b)) / 2;         // src/coreLoop.cpp:56
}

// ------------------------------------------------------------ reductions ----
// red[j] = sum over tiles of rowA + rowGB (fixed order) = rowSums(Z), R/update_vb.R:179
// rowGB holds gb_rows rows per tile (the generic kernel splits a tile over several workgroups), rowA one.
// Workgroup = 64 predictors x 4 interleaved quarters of the tiles (one thread per predictor left the GPU three quarters empty and
// every thread waiting on its own 625 loads: 0.27 ms at C3); the four partial sums are added in fixed order.
__global__ __launch_bounds__(256) void aq_k_reduce_rows(const double *__restrict__ rowA, const double *__restrict__ rowGB,
                                                        double *__restrict__ red, int ntile, int p_pad, int gb_rows) {
  __shared__ double part[4][64];
  const int jl = threadIdx.x & 63, c = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  double s = 0.0;
  if (j < p_pad) {
    for (int t = c; t < ntile; t += 4) {
      double gb = rowGB[(size_t)t * gb_rows * p_pad + j];
      for (int r = 1; r < gb_rows; r++) gb += rowGB[((size_t)t * gb_rows + r) * p_pad + j];
      s += (rowA ? rowA[(size_t)t * p_pad + j] : 0.0) + gb;   // rowA == NULL: the sweep kernel already folded a into rowGB
    }
  }
  part[c][jl] = s;
  __syncthreads();
  if (c == 0 && j < p_pad) red[j] = ((part[0][jl] + part[1][jl]) + part[2][jl]) + part[3][jl];
}

// sums[0] <- column sums added over the chained SNP-segment slots (fixed order); ||R||^2 is the last segment's.
__global__ void aq_k_combine_segment_sums(double *sums, int q_pad, int nslot) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= q_pad) return;
  size_t Q = q_pad;
  for (int v = 0; v < 4; v++) {
    double acc = sums[v * Q + k];
    for (int s = 1; s < nslot; s++) acc += sums[(size_t)s * 5 * Q + v * Q + k];
    sums[v * Q + k] = acc;
  }
  sums[4 * Q + k] = sums[(size_t)(nslot - 1) * 5 * Q + 4 * Q + k];
}

__device__ __forceinline__ double aq_block_sum_1024(double v, double *sh) {
  int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

// one workgroup: red[p_pad+0] = sum gam, +1 = sum_k tau_k colSums(m2)_k, +2 = sum zeta (before its update)
__global__ void aq_k_reduce_q_scalars(AqQvec v, double *red_tail) {
  __shared__ double sh[1024];
  double a = 0, b = 0, z = 0;
  for (int k = threadIdx.x; k < v.q; k += blockDim.x) {
    a += v.sums[k];
    b += v.tau[k] * v.sums[(size_t)v.q_pad + k];
    z += v.zeta[k];
  }
  a = aq_block_sum_1024(a, sh);
  b = aq_block_sum_1024(b, sh);
  z = aq_block_sum_1024(z, sh);
  if (threadIdx.x == 0) {
    red_tail[0] = a;
    red_tail[1] = b;
    red_tail[2] = z;
    for (int i = 3; i < AQ_RED_EXTRA; i++) red_tail[i] = 0.0;
  }
}

__global__ void aq_k_take_reduced_scalars(AqScalars *sc, const double *red_tail) {
  sc->S_gam = red_tail[0];
  sc->T2 = red_tail[1];
  sc->sum_zeta_old = red_tail[2];
}

// ------------------------------------------------- p-vector: S12-S18 --------
struct AqPvec {
  double *theta, *sig2_theta, *L, *lam2_inv, *Q;
  const double *rsZ;     // all-reduced row sums of Z
  double *part;          // [3][nblk] partial sums: theta, lam*shr*(...), sig2_theta
  int p, p_pad;
  double shr, m0, A2_inv, df;
};

// S12 + the shared stopping rule of Q_approx_vec (R/utils.R:380-423, note N2): bit i of the
// AND-mask is set iff every x > 1 element has |Delta - 1| < eps2 at iteration counter j = i + 2.
__global__ void aq_k_pvec_L(AqPvec v, AqScalars *sc, double c_s, int annealing) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long m0 = ~0ull, m1 = ~0ull;
  if (j < v.p) {
    double th = v.theta[j];
    double L = c_s * sc->sig02_inv * v.shr * (th * th + v.sig2_theta[j] - 2 * th * v.m0 + v.m0 * v.m0) / 2 / v.df;   // :241
    v.L[j] = L;
    if (!annealing && L > 1.0) {
      AqLentz s;
      aq_lentz_init(&s);
      m0 = 0; m1 = 0;
      for (int it = 0; it < 128; it++) {
        double d = aq_lentz_step(&s, L, it + 2);
        if (d < 1e-7) {
          if (it < 64) m0 |= (1ull << it); else m1 |= (1ull << (it - 64));
        }
      }
    }
  }
  // wave-level AND, then one atomic per wave
  for (int o = 32; o > 0; o >>= 1) {
    m0 &= __shfl_xor(m0, o, 64);
    m1 &= __shfl_xor(m1, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && !annealing) {
    if (m0 != ~0ull) atomicAnd(&sc->lentz_mask[0], m0);
    if (m1 != ~0ull) atomicAnd(&sc->lentz_mask[1], m1);
  }
}

__global__ void aq_k_reset_lentz(AqScalars *sc) {
  sc->lentz_mask[0] = ~0ull;
  sc->lentz_mask[1] = ~0ull;
  sc->lentz_iters = 0;
}

// S14, S16, S17 and the partial sums S18/S19 need.  R/atlasqtl_global_local_core.R:244-288.
__global__ void aq_k_pvec_finish(AqPvec v, AqScalars *sc, double c, double c_s, int annealing, int q_total) {
  __shared__ double sh[256];
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  double th_new = 0, t2 = 0, s2t = 0;
  // shared Lentz iteration count: first set bit (iteration counter j = bit + 2)
  int nit = 0;
  bool any_upper = false;
  if (!annealing) {
    unsigned long long a0 = sc->lentz_mask[0], a1 = sc->lentz_mask[1];
    if (a0 == ~0ull && a1 == ~0ull) {
      nit = 0;   // no element above 1 (or converged at once everywhere): handled per element below
      any_upper = false;
    } else {
      any_upper = true;
      nit = a0 ? (__ffsll((long long)a0)) : (a1 ? 64 + __ffsll((long long)a1) : 129);
    }
    if (j == 0) sc->lentz_iters = nit;
  }
  if (j < v.p) {
    double L = v.L[j];
    double lam;
    if (annealing && v.df != 1.0) {
      lam = aq_annealed_lam2_inv_df(L, c_s, v.df);                                   // R/update_vb.R:76-81 (Kummer's 1F1)
      v.Q[j] = 0.0;
    } else if (annealing) {
      // gsl::gamma_inc(2-c, L) / (gsl::gamma_inc(1-c, L) L) - 1, with Gamma(a+1,x) = a Gamma(a,x) + x^a e^-x
      double aa = 1.0 - c_s;
      double ga = aq_gamma_inc_upper(aa, L);
      double g1 = aa * ga + exp(aa * log(L) - L);
      lam = g1 / (ga * L) - 1.0;                                                     // R/update_vb.R:74
      v.Q[j] = 0.0;
    } else {
      double Q;
      if (L <= 1.0) {
        Q = aq_expint_e1_small(L) * exp(L);                                          // R/utils.R:387
      } else {
        AqLentz s;
        aq_lentz_init(&s);
        int n_it = any_upper ? nit : 1;
        for (int it = 0; it < n_it; it++) aq_lentz_step(&s, L, it + 2);
        Q = aq_lentz_finish(&s, L);                                                  // R/utils.R:419
      }
      v.Q[j] = Q;
      if (v.df == 3.0) {                                                             // :258 (L is L / df here, :241)
        lam = exp(-1.0986122886681098 /* log 3 */ - log(L) + log(1.0 - L * Q) - log(Q * (1.0 + L) - 1.0)) - 1.0 / 3.0;
      } else if (v.df > 3.0) {                                                       // :260-272 (df = 5, 7)
        const int ex = ((int)v.df + 1) / 2;
        lam = exp(log(aq_hs_integral(v.df, L * v.df, ex, ex, Q)) - log(aq_hs_integral(v.df, L * v.df, ex, ex - 1, Q)));
      } else
        lam = 1.0 / (Q * L) - 1.0;                                                   // :254
    }
    v.lam2_inv[j] = lam;
    double s02 = sc->sig02_inv * lam * v.shr;
    double sig2_theta = 1.0 / (c * ((double)q_total + s02));                         // update_sig2_c0_vb_, :278
    th_new = c * sig2_theta * (v.rsZ[j] + s02 * v.m0 - sc->sum_zeta_old);            // update_theta_vb_, :280
    v.sig2_theta[j] = sig2_theta;
    v.theta[j] = th_new;
    t2 = lam * v.shr * (th_new * th_new + sig2_theta - 2 * th_new * v.m0 + v.m0 * v.m0);   // :285-286
    s2t = sig2_theta;
  }
  int nblk = gridDim.x;
  double r;
  sh[threadIdx.x] = th_new; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[blockIdx.x] = r;
  sh[threadIdx.x] = t2; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[nblk + blockIdx.x] = r;
  sh[threadIdx.x] = s2t; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[2 * nblk + blockIdx.x] = r;
}

// S13, S15, S18 (one workgroup).  R/atlasqtl_global_local_core.R:242,276,283-288.
__global__ void aq_k_scalars_post(AqPvec v, AqScalars *sc, double c_s, int nblk) {
  __shared__ double sh[1024];
  double a = 0, b = 0, d = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    a += v.part[i];
    b += v.part[nblk + i];
    d += v.part[2 * nblk + i];
  }
  a = aq_block_sum_1024(a, sh);
  b = aq_block_sum_1024(b, sh);
  d = aq_block_sum_1024(d, sh);
  if (threadIdx.x == 0) {
    double rho_xi_inv = c_s * (v.A2_inv + sc->sig02_inv);         // :242 (old sig02_inv)
    double xi_inv = 1.0 / rho_xi_inv;                              // :276 (nu_xi_inv_vb = 1)
    double nu_s0 = c_s * (0.5 + (double)v.p / 2) - c_s + 1;        // :283
    double rho_s0 = c_s * (xi_inv + b / 2);                        // :285
    sc->rho_xi_inv = rho_xi_inv;
    sc->xi_inv = xi_inv;
    sc->nu_s0 = nu_s0;
    sc->rho_s0 = rho_s0;
    sc->sig02_inv = nu_s0 / rho_s0;                                // :288
    sc->sum_theta = a;
    sc->sum_sig2_theta = d;
  }
}

// ---- the global-only core (atlasqtl_global_core_, R/atlasqtl_global_core.R:238-256): one scale for all hotspot propensities
// sig2_theta (a scalar in the reference, stored per predictor here), theta, and the partial sums of theta and of
// sig2_theta + theta^2 - 2 theta m0 + m0^2
__global__ void aq_k_pvec_global(AqPvec v, AqScalars *sc, double c, int q_total) {
  __shared__ double sh[256];
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  double th_new = 0, t2 = 0, s2t = 0;
  if (j < v.p) {
    const double s02 = sc->sig02_inv * v.shr;
    const double sig2_theta = 1.0 / (c * ((double)q_total + s02));                  // update_sig2_c0_vb_(q, 1 / sig02_inv / shr, c), :240
    th_new = c * sig2_theta * (v.rsZ[j] + s02 * v.m0 - sc->sum_zeta_old);           // update_theta_vb_, :242
    v.sig2_theta[j] = sig2_theta;
    v.theta[j] = th_new;
    v.lam2_inv[j] = 1.0;
    t2 = sig2_theta + th_new * th_new - 2 * th_new * v.m0 + v.m0 * v.m0;            // :253
    s2t = sig2_theta;
  }
  int nblk = gridDim.x;
  double r;
  sh[threadIdx.x] = th_new; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[blockIdx.x] = r;
  sh[threadIdx.x] = t2; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[nblk + blockIdx.x] = r;
  sh[threadIdx.x] = s2t; __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  r = sh[0]; __syncthreads();
  if (threadIdx.x == 0) v.part[2 * nblk + blockIdx.x] = r;
}
// nu_s0_vb, rho_s0_vb, sig02_inv_vb of the global-only core (nu_s0 = rho_s0 = 1/2, :96, :252-255)
__global__ void aq_k_scalars_post_global(AqPvec v, AqScalars *sc, double c_s, int nblk) {
  __shared__ double sh[1024];
  double a = 0, b = 0, d = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    a += v.part[i];
    b += v.part[nblk + i];
    d += v.part[2 * nblk + i];
  }
  a = aq_block_sum_1024(a, sh);
  b = aq_block_sum_1024(b, sh);
  d = aq_block_sum_1024(d, sh);
  if (threadIdx.x == 0) {
    double nu_s0 = c_s * (0.5 + (double)v.p / 2) - c_s + 1;        // :252
    double rho_s0 = c_s * (0.5 + b / 2);                           // :253
    sc->nu_s0 = nu_s0;
    sc->rho_s0 = rho_s0;
    sc->sig02_inv = nu_s0 / rho_s0;                                // :255
    sc->sum_theta = a;
    sc->sum_sig2_theta = d;
    sc->sum_theta_sq = b - d;                                      // sum (theta - m0)^2, for e_theta_
  }
}
// e_theta_ (R/elbo.R:74-81) with vec_sum_log_det_theta of R/atlasqtl_global_core.R:397 -> sc->elbo_C.  One thread.
__global__ void aq_k_elbo_C_global(AqPvec v, AqScalars *sc) {
  const double p = (double)v.p;
  const double log_sig02_inv = aq_digamma(sc->nu_s0) - log(sc->rho_s0);
  const double s2t = sc->sum_sig2_theta / p;
  const double vsld = p * (log_sig02_inv + log(v.shr) + log(s2t));
  const double sig02 = v.shr * sc->sig02_inv;
  sc->elbo_C = (vsld - sig02 * sc->sum_theta_sq - p * sig02 * s2t + p) / 2;
}

// S19: zeta.  R/update_vb.R:99-110 (is_mat = FALSE).
__global__ void aq_k_qpost(AqQvec v, const AqScalars *sc, double c, double sig2_zeta, double t02_inv) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= v.q_pad) return;
  if (k >= v.q) { v.zeta[k] = 0.0; return; }
  double sZ = v.sums[(size_t)3 * v.q_pad + k];                 // colSums(Z) = colSums(a) + colSums(gam*b)
  for (int ch = 0; ch < v.nchunk; ch++) sZ += v.colApart[(size_t)ch * v.q_pad + k];
  v.zeta[k] = c * sig2_zeta * (sZ + t02_inv * v.n0[k] - sc->sum_theta);
}

// ------------------------------------------------------------------ ELBO ----
// e_theta_hs_ (R/elbo.R:85-92, df = 1), replicated p-sum -> sc->elbo_C.  One workgroup.
__global__ void aq_k_elbo_C(AqPvec v, AqScalars *sc) {
  __shared__ double sh[1024];
  double log_sig02p = aq_digamma(sc->nu_s0) - log(sc->rho_s0) + log(v.shr);   // log_sig02_inv_vb + log(shr_fac_inv), :477
  double sig02p = sc->sig02_inv * v.shr;
  double acc = 0.0;
  for (int j = threadIdx.x; j < v.p; j += blockDim.x) {
    double th = v.theta[j], s2 = v.sig2_theta[j], lam = v.lam2_inv[j], L = v.L[j];
    const double quad = log_sig02p / 2 - sig02p * lam * (th * th + s2 - 2 * v.m0 * th + v.m0 * v.m0) / 2 + (log(s2) + 1) / 2;
    if (v.df == 3.0) {   // R/elbo.R:95-105: log(6) + log(3)/2 - log(pi) - log_B + df L lam + ..., log_B = log(9) - log(Q (1 + L) - 1)
      acc += quad + 1.791759469228055 + 0.5493061443340549 - 1.1447298858494001741434273513531
             - (2.1972245773362196 - log(v.Q[j] * (1.0 + L) - 1.0)) + 3.0 * L * lam;
    } else if (v.df > 3.0) {   // R/elbo.R:107-124: -log(pi)/2 - lgamma(df/2) + df log(df)/2 + lfactorial((df-1)/2) - log_B + df L lam + ...
      const int ex = ((int)v.df + 1) / 2;
      const double log_B = -log(aq_hs_integral(v.df, L * v.df, ex, ex - 1, v.Q[j]));
      acc += quad - 0.5 * 1.1447298858494001741434273513531 - lgamma(v.df / 2) + v.df * log(v.df) / 2 + lgamma((v.df - 1) / 2 + 1.0)
             - log_B + v.df * L * lam;
    } else
      acc += quad - 1.1447298858494001741434273513531 /* log(pi) */ + L * lam + log(v.Q[j]);
  }
  acc = aq_block_sum_1024(acc, sh);
  if (threadIdx.x == 0) sc->elbo_C = acc;
}

// local (own traits) ELBO sums -> ered[0..7].  One workgroup.
//  0 Hsum  1 sum log_tau_e*sg  2 sum sg*(log sig2b + 1)  3 A_loc (e_y_)  4 E_loc (e_tau_)  5 sum (zeta-n0)^2
// ELBO-local eta/kappa/log_tau use c = 1 (R/atlasqtl_global_local_core.R:456-464, note N4).
__global__ void aq_k_elbo_q(AqQvec v, const AqScalars *sc, const double *Hpart, int nHpart, double *ered) {
  __shared__ double sh[1024];
  double h = 0, s1 = 0, s2 = 0, A = 0, E = 0, D = 0;
  for (int i = threadIdx.x; i < nHpart; i += blockDim.x) h += Hpart[i];
  double sig2_inv = sc->sig2_inv;
  double nm1 = (double)(v.n - 1);
  size_t Q = v.q_pad;
  for (int k = threadIdx.x; k < v.q; k += blockDim.x) {
    double sg = v.sums[k], sm2 = v.sums[Q + k], sb2 = v.sums[2 * Q + k], rn = v.sums[4 * Q + k];
    double eta_e = v.eta_h[k] + v.nobs[k] / 2 + sg / 2;
    double kappa_e = v.na ? v.kappa_h[k] + (rn + sig2_inv * sm2 + sb2) / 2
                          : v.kappa_h[k] + (rn + (nm1 + sig2_inv) * sm2 - nm1 * sb2) / 2;
    double log_tau_e = aq_digamma(eta_e) - log(kappa_e);
    double tau = v.tau[k];
    s1 += log_tau_e * sg;
    s2 += v.na ? (v.sums[5 * Q + k] + sg) : sg * (log(v.sig2b[k]) + 1);   // sum_j gam_jk (log sig2_beta + 1), R/elbo.R:28-32
    A += v.nobs[k] * (log_tau_e - 1.8378770664093454835606594728112 /* log(2 pi) */) / 2
         - tau * (kappa_e - sm2 * sig2_inv / 2 - v.kappa_h[k]);                                   // e_y_, R/elbo.R:135-146
    E += (v.eta_h[k] - eta_e) * log_tau_e - (v.kappa_h[k] - kappa_e) * tau + v.eta_h[k] * log(v.kappa_h[k])
         - eta_e * log(kappa_e) - lgamma(v.eta_h[k]) + lgamma(eta_e);                             // e_tau_, R/elbo.R:63-68
    double dz = v.zeta[k] - v.n0[k];
    D += dz * dz;
  }
  h = aq_block_sum_1024(h, sh);
  s1 = aq_block_sum_1024(s1, sh);
  s2 = aq_block_sum_1024(s2, sh);
  A = aq_block_sum_1024(A, sh);
  E = aq_block_sum_1024(E, sh);
  D = aq_block_sum_1024(D, sh);
  if (threadIdx.x == 0) {
    ered[0] = h; ered[1] = s1; ered[2] = s2; ered[3] = A; ered[4] = E; ered[5] = D; ered[6] = 0; ered[7] = 0;
  }
}

struct AqElboConst {
  double nu_h, rho_h, A2_inv, t02_inv, vec_sum_log_det_zeta, sig2_zeta;
  double p, q_total;
  int global_only;   // atlasqtl_global_core_: no local scales, sig02_inv ~ Gamma(1/2, 1/2) (R/atlasqtl_global_core.R:96,413-416)
};

__device__ __forceinline__ double aq_e_sig2_inv(double nu, double nu_vb, double log_s, double rho, double rho_vb,
                                                double s) {   // R/elbo.R:41-46
  return (nu - nu_vb) * log_s - (rho - rho_vb) * s + nu * log(rho) - nu_vb * log(rho_vb) - lgamma(nu) + lgamma(nu_vb);
}

// assemble the scalar ELBO from the all-reduced sums.  R/atlasqtl_global_local_core.R:440-495.
__global__ void aq_k_elbo_final(AqScalars *sc, const double *ered, AqElboConst k) {
  double S_gam = sc->S_gam, T2 = sc->T2;
  double nu_e = k.nu_h + S_gam / 2;                    // update_nu_vb_ (c = 1)
  double rho_e = k.rho_h + T2 / 2;                     // update_rho_vb_
  double log_sig2_inv_e = aq_digamma(nu_e) - log(rho_e);
  double log_sig02_inv = aq_digamma(sc->nu_s0) - log(sc->rho_s0);
  double log_xi_inv = aq_digamma(1.0) - log(sc->rho_xi_inv);
  double sig2_inv = sc->sig2_inv;
  double A = ered[3];
  double B = log_sig2_inv_e * S_gam / 2 + ered[1] / 2 - sig2_inv * T2 / 2 + ered[0]
             - k.p * k.q_total * k.sig2_zeta / 2 - k.q_total * sc->sum_sig2_theta / 2 + ered[2] / 2;   // R/elbo.R:10-34
  double C = sc->elbo_C;
  double D = (k.vec_sum_log_det_zeta - k.t02_inv * ered[5] - k.q_total * k.t02_inv * k.sig2_zeta + k.q_total) / 2;  // R/elbo.R:153-161
  double E = ered[4];
  double F = -0.5 * log_sig02_inv - sc->xi_inv * sc->sig02_inv + log_xi_inv / 2 - lgamma(0.5)
             - (sc->nu_s0 - 1) * log_sig02_inv + sc->rho_s0 * sc->sig02_inv - sc->nu_s0 * log(sc->rho_s0)
             + lgamma(sc->nu_s0);                                                                       // R/elbo.R:49-56
  double G = aq_e_sig2_inv(0.5, 1.0, log_xi_inv, k.A2_inv, sc->rho_xi_inv, sc->xi_inv);
  double H = aq_e_sig2_inv(k.nu_h, nu_e, log_sig2_inv_e, k.rho_h, rho_e, sig2_inv);
  if (k.global_only) {   // elbo_F + elbo_G of R/atlasqtl_global_core.R:413-416 take the place of the horseshoe's F, G, H
    F = 0.0;
    G = aq_e_sig2_inv(0.5, sc->nu_s0, log_sig02_inv, 0.5, sc->rho_s0, sc->sig02_inv);
  }
  sc->elbo = A + B + C + D + E + F + G + H;
}
