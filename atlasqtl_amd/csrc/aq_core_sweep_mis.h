// aq_core_sweep_mis.h -- the blocked f64-MFMA sweep for Y with missing values
// (reference coreDualMisLoop, src/coreLoop.cpp:91-138, called from R/atlasqtl_global_local_core.R:172).
//
// With mis_pat (n x q, 1 = observed) the reference's Gram matrix becomes trait specific, cp_X - cp_X_rm[[k]]
// (R/atlasqtl_global_local_core.R:27-31).  In n-space that is a masked residual
//     R_ik = mis_ik (y_ik - sum_j x_ij beta_jk)
// and the blocked Gauss-Seidel of aq_core_sweep.h carries over with two changes:
//   * S = X_b' R_K is unchanged (R is already masked); the update becomes R_K -= mis .* (X_b delta): the MFMA
//     result is re-masked in registers (the mask of a wave's residual tiles is 4 NT bits per lane);
//   * the in-block coupling uses the trait's own Gram block
//         G_b^(k) = X_b' diag(mis_k) X_b = X_b'X_b - Xm_k' Xm_k,      Xm_k = the rows of X_b at trait k's missing samples,
//     a rank-m_k correction computed with the SAME f64 MFMA: for the 16x16x4 instruction the A operand (16 SNPs x 4
//     samples) and the B operand (4 samples x 16 SNPs) of Xm_k'Xm_k are the same register -- lane l holds
//     x[I_k[4t + (l>>4)]][l & 15], one gathered 128-byte row segment per 16 lanes.  Its diagonal is X_norm_sq(j,k)
//     (R/atlasqtl_global_local_core.R:23), from which the p x q sig2_beta_vb (R/update_vb.R:45) follows per entry.
//     The 16 corrections of block b+1 are computed by the six waves that would otherwise idle while wave 0 runs the
//     sequential pass of block b.
// n beyond one workgroup's registers (8 waves x 16 tiles = 2048 samples): C workgroups share a trait tile, each
// holds n_pad/C samples.  Per SNP block every one of them publishes its 16x16 partial S (agent-scope atomic stores,
// then a flag), waits for the C-1 others (bounded spin), adds the partials in the same fixed order and runs the SAME sequential pass
// redundantly -- so delta is bitwise identical in all of them and no second exchange is needed.  Workgroups of a
// tile are adjacent in dispatch order, hence a waiting one always has its partners resident or next to be dispatched.
// Everything per (j,k) uses the same trait-tiled arrays and the same 6 column sums as aq_trait_wave.h (the generic
// kernel, which stays the fallback for more than AQ_MIS_MMAX missing samples in a trait and for n > 16384).
#pragma once
#include <hip/hip_runtime.h>
#include "aq_special.h"
#include "aq_core_sweep.h"
#include "aq_vec_kernels.h"
#include "aq_core_sweep_la.h"   // aq_static_for, aq_row16_sum

#define AQ_MIS_MMAX 1024  // most missing samples of one trait the LDS index lists hold (16-bit indices, padded to 16)

struct AqMisArgs {
  const double2 *XA, *XU;  // MFMA operand layouts of X (aq_core_sweep.h)
  const double *G;         // [nb][16][16] X_b'X_b
  const double *XR;        // [nb][NR][16] X_b row-major (sample, snp); rows >= n are zero, row n_pad is the padding target
  double *R;               // [ntile][n_pad][16] masked residual
  const double *mis;       // [ntile][n_pad][16] 1 = observed
  double *gam, *mu;        // [ntile][p_pad][16]
  const double *Aarr, *Barr;
  const double *tau, *log_tau, *sig2b;   // [q_pad]; sig2b = initial sig2_beta_vb (init mode only)
  const AqScalars *sc;     // sig2_inv, log_sig2_inv of this sweep
  const int *midx;         // [ntile][16][Mmax] sample indices of the missing entries of each trait, padded with n_pad
  const int *mcnt4;        // [ntile][16] number of 4-sample groups in each list (a multiple of 4: lists are padded to 16 samples)
  double *sums;            // [6][q_pad]: sum gam, sum m2, sum X_norm_sq (m2 - beta^2), sum gam*b, ||R||^2, sum gam*log sig2_beta
  double *rowGB;           // [ntile][p_pad]
  double c;
  int p, q, p_pad, q_pad, n_pad, nb, ntile, dmode, mode, NR, Mmax;
  int C;                   // workgroups per trait tile (sample split); block tile*C + part
  double *Pbuf;            // [ntile][2][C][256] partial S of each part, double-buffered by block parity
  int *pflag;              // [ntile][C] number of blocks whose partial S this part has published
  int *errflag;            // set when a bounded wait expires
  double *rnpart;          // [C][q_pad] partial ||R_k||^2 of each part (C > 1)
  int nseg;                // > 1 (C == 1 only): chained SNP segments as in aq_core_sweep_la.h, block s*ntile + k = segment s of tile k;
  int *done;               //   done[tile] = segments of that tile already finished; sums land in slot s of 6 rows each
};

// NT: 16-sample residual tiles per wave; 8 waves, C parts: n_pad = 128 NT C.
template <int NT>
__global__ __launch_bounds__(512, 1) void aq_core_sweep_mis_kernel(const AqMisArgs a) {
  constexpr int NW = 8;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  int tile_ = blockIdx.x / a.C, seg = 0, seg_b0 = 0, seg_b1 = a.nb;
  const int part = blockIdx.x - tile_ * a.C;
  if (a.nseg > 1) {
    // chained SNP segments (more trait tiles than CUs): this workgroup continues tile k where segment s-1 left its residual
    seg = blockIdx.x / a.ntile;
    tile_ = blockIdx.x - seg * a.ntile;
    seg_b0 = (int)((long long)a.nb * seg / a.nseg);
    seg_b1 = (int)((long long)a.nb * (seg + 1) / a.nseg);
    if (seg > 0) {
      if (threadIdx.x == 0) {
        int tries = 0;
        while (__hip_atomic_load(&a.done[tile_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg) {
          __builtin_amdgcn_s_sleep(32);
          if (++tries > 4000000) { *a.errflag = 1; break; }   // bounded: never hang the GPU
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
  }
  const int tile = tile_;
  const bool lead = (part == 0);          // the part that records the tile's results
  const int NTT = NT * NW * a.C;          // residual tiles of the whole sample axis
  const int wt0 = (part * NW + w) * NT;   // first residual tile of this wave
  const bool helper = tid < 256;
  const int hj = (tid >> 4) & 15, hk = tid & 15;

  extern __shared__ double lds[];
  double *Sp = lds;                       // [NW][256] per-wave partial S [snp][trait]
  double *LcA = Sp + NW * 256;            // c (A - log(sig2_beta)/2 + cst)      src/coreLoop.cpp:127-129
  double *Lm1 = LcA + 256, *LB = Lm1 + 256, *Lgam = LB + 256, *Lmu = Lgam + 256, *Ldel = Lmu + 256;
  double *Lcoef = Ldel + 256;             // c sig2_beta tau                      :125
  double *Lci2s = Lcoef + 256;            // K = coef^2 c / (2 sig2_beta)
  double *Ls2 = Lci2s + 256, *Lls2 = Ls2 + 256, *Lxn = Lls2 + 256;
  double *Lred = Lxn + 256;               // [5][256] running column sums per helper thread
  double *Lrn = Lred + 5 * 256;           // [NW*4][16]
  double *LGk = Lrn + NW * 4 * 16;        // [2][256*17]: G_b^(k)[i][j] at (i*16 + j)*17 + k
  int *Lcnt = (int *)(LGk + 2 * 256 * 17);            // [16]
  unsigned short *Lidx = (unsigned short *)(Lcnt + 16);   // [16][Mmax]

  // ---- residual tiles and their mask bits into registers ----
  // (f64 MFMA D layout row = 4 reg + (lane >> 4): the host refuses to run this kernel on a device that reports the other
  // map; with it fixed, every tile address below is the wave's base plus a compile-time offset)
  aq_d4 Rr[NT];
  unsigned long long mb = 0;
  {
    const double *Rg = a.R + (size_t)tile * a.n_pad * 16 + (size_t)(16 * wt0 + g) * 16 + col;
    const double *Mg = a.mis + (size_t)tile * a.n_pad * 16 + (size_t)(16 * wt0 + g) * 16 + col;
    aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        Rr[t][r] = Rg[(16 * t + 4 * r) * 16];
        if (Mg[(16 * t + 4 * r) * 16] != 0.0) mb |= 1ull << (4 * t + r);
      }
    });
  }
  auto remask = [&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
#pragma unroll
    for (int r = 0; r < 4; r++) Rr[t][r] = ((mb >> (4 * t + r)) & 1ull) ? Rr[t][r] : 0.0;
  };
  for (int e = tid; e < 16 * a.Mmax; e += 512) Lidx[e] = (unsigned short)a.midx[(size_t)tile * 16 * a.Mmax + e];
  if (tid < 16) Lcnt[tid] = a.mcnt4[tile * 16 + tid];
  if (helper)
#pragma unroll
    for (int v = 0; v < 5; v++) Lred[v * 256 + tid] = 0.0;

  const double2 *XAw = a.XA + (size_t)wt0 * 128 + lane;   // + (b*NTT + t)*128 + h*64
  const double2 *XUw = a.XU + (size_t)wt0 * 128 + lane;
  const int kk_h = tile * 16 + hk;
  const bool kvalid = kk_h < a.q;
  const size_t tbase = (size_t)tile * a.p_pad * 16;
  const double tau_h = a.tau[kk_h];
  const double s2init_h = a.sig2b[kk_h];
  const double cstna_h = -(a.log_tau[kk_h] + a.sc->log_sig2_inv) / 2;            // src/coreLoop.cpp:108
  const double sig2_inv = a.sc->sig2_inv;
  __syncthreads();

  // G_bb^(k) for the 16 traits of the tile -> LGk[buf]; waves 1,2,3,5,6,7 (never on the recurrence wave's SIMD)
  auto compute_gk = [&](int bb, int buf) __attribute__((always_inline)) {
    if (w == 0 || w == 4) return;
    const int slot = w < 4 ? w - 1 : w - 2;
    const double *xr = a.XR + (size_t)bb * a.NR * 16 + col;
    double gb[4];
#pragma unroll
    for (int r = 0; r < 4; r++) gb[r] = a.G[(size_t)bb * 256 + (4 * r + g) * 16 + col];
    double *dst = LGk + buf * (256 * 17);
    for (int k = slot; k < 16; k += 6) {
      aq_d4 acc = (aq_d4){0, 0, 0, 0};
      const int n4 = Lcnt[k];
      const unsigned short *ix = Lidx + k * a.Mmax + g;
      for (int t = 0; t < n4; t += 4) {   // lists are padded to whole groups of 16 samples: 4 gathers in flight per step
        const int i0 = ix[4 * t], i1 = ix[4 * t + 4], i2 = ix[4 * t + 8], i3 = ix[4 * t + 12];
        const double x0 = xr[(size_t)i0 * 16], x1 = xr[(size_t)i1 * 16], x2 = xr[(size_t)i2 * 16], x3 = xr[(size_t)i3 * 16];
        acc = aq_mfma(x0, x0, acc);
        acc = aq_mfma(x1, x1, acc);
        acc = aq_mfma(x2, x2, acc);
        acc = aq_mfma(x3, x3, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) dst[((4 * r + g) * 16 + col) * 17 + k] = gb[r] - acc[r];
    }
  };

  if (a.mode == 1) {
    // ---------------- init: R = mis .* (Y - X (gam*mu)), initial column sums ----------------
    for (int b = 0; b < a.nb; b++) {
      compute_gk(b, 0);
      double gm = 0.0, mu = 0.0;
      if (helper) {
        size_t off = tbase + (size_t)(16 * b) * 16 + tid;
        gm = a.gam[off];
        mu = a.mu[off];
        Ldel[tid] = gm * mu;                                  // update_beta_vb_, R/update_vb.R:17
      }
      __syncthreads();
      if (helper && kvalid && (16 * b + hj) < a.p) {
        double xn = LGk[(hj * 16 + hj) * 17 + hk];            // X_norm_sq(j,k)
        double be = gm * mu;
        double m2 = (mu * mu + s2init_h) * gm;                // first m2_beta uses the initial q-vector sig2_beta_vb, :113
        Lred[tid] += gm;
        Lred[256 + tid] += m2;
        Lred[512 + tid] += xn * (m2 - be * be);
      }
      double nd[4];
#pragma unroll
      for (int s = 0; s < 4; s++) nd[s] = -Ldel[(4 * s + g) * 16 + col];
      const double2 *xu = XUw + (size_t)b * NTT * 128;
      aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        double2 u0 = xu[t * 128], u1 = xu[t * 128 + 64];
        Rr[t] = aq_mfma(u0.x, nd[0], Rr[t]);
        Rr[t] = aq_mfma(u0.y, nd[1], Rr[t]);
        Rr[t] = aq_mfma(u1.x, nd[2], Rr[t]);
        Rr[t] = aq_mfma(u1.y, nd[3], Rr[t]);
        __builtin_amdgcn_sched_barrier(0);
      });
      __syncthreads();
    }
    aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) { remask(tc); });
  } else {
    // ---------------- full sweep ----------------
    double st_A = 0, st_g = 0, st_m = 0, st_B = 0;
    auto stage_load = [&](int b) __attribute__((always_inline)) {
      size_t off = tbase + (size_t)(16 * b) * 16 + tid;
      st_A = a.Aarr[off];
      st_g = a.gam[off];
      st_m = a.mu[off];
      st_B = a.Barr[off];
    };
    auto stage_commit = [&](int buf) __attribute__((always_inline)) {
      const double xn = LGk[buf * (256 * 17) + (hj * 16 + hj) * 17 + hk];
      const double s2 = 1.0 / (a.c * (xn + sig2_inv) * tau_h);            // update_sig2_beta_vb_, R/update_vb.R:45
      const double ls2 = log(s2);
      LcA[tid] = a.c * (st_A - 0.5 * ls2 + cstna_h);
      Lcoef[tid] = a.c * s2 * tau_h;
      Lci2s[tid] = (a.c * s2 * tau_h) * (a.c * s2 * tau_h) * (a.c * 0.5 / s2);   // coef^2 c/(2 sig2_beta): x = cA - s^2 K
      Ls2[tid] = s2;
      Lls2[tid] = ls2;
      Lxn[tid] = xn;
      Lm1[tid] = st_g * st_m;
      LB[tid] = st_B;
    };
    // prologue: S of block 0, G^(k) of block 0, staged scalars of block 0
    aq_d4 acc = (aq_d4){0, 0, 0, 0};
    aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      const double2 *xa0 = XAw + (size_t)seg_b0 * NTT * 128;
      double2 a0 = xa0[t * 128], a1 = xa0[t * 128 + 64];
      acc = aq_mfma(a0.x, Rr[t][0], acc);
      acc = aq_mfma(a0.y, Rr[t][1], acc);
      acc = aq_mfma(a1.x, Rr[t][2], acc);
      acc = aq_mfma(a1.y, Rr[t][3], acc);
      __builtin_amdgcn_sched_barrier(0);
    });
    compute_gk(seg_b0, seg_b0 & 1);
    if (helper) stage_load(seg_b0);
    __syncthreads();
    if (helper) stage_commit(seg_b0 & 1);

    bool dead = false;   // a bounded wait on a partner expired (reported through errflag)
    for (int b = seg_b0; b < seg_b1; b++) {
      const bool more = (b + 1 < seg_b1);
      const int buf = b & 1;
      if (helper && more) stage_load(b + 1);
#pragma unroll
      for (int i = 0; i < 4; i++) Sp[w * 256 + (4 * i + g) * 16 + col] = acc[i];
      aq_lds_barrier();

      if (w == 0) {
        if (a.C > 1) {
          // ---- sample split: publish this part's partial S, collect the others', add in fixed order ----
          // All exchanged words are agent-scope atomics (sc1: written through to, and read from, the level the XCDs
          // share), ordered by program order + s_waitcnt.  No release/acquire fence: at agent scope it writes back and
          // invalidates this XCD's whole L2, where the X operand panels live.  The whole wave takes part (lane = 4
          // entries of the 16x16 block) so that every load of a step is in flight at once.
          double own[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int e = lane + 64 * r;
            double v = Sp[e];
#pragma unroll
            for (int ww = 1; ww < NW; ww++) v += Sp[ww * 256 + e];
            own[r] = v;
          }
          double *slot = a.Pbuf + ((size_t)(tile * 2 + buf) * a.C) * 256 + lane;
#pragma unroll
          for (int r = 0; r < 4; r++)
            __hip_atomic_store(&slot[(size_t)part * 256 + 64 * r], own[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the partial is performed before the flag goes up
          if (lane == 0) __hip_atomic_store(&a.pflag[tile * a.C + part], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while (!dead) {   // lane c polls part c's flag
            int f = b + 1;
            if (lane < a.C && lane != part) f = __hip_atomic_load(&a.pflag[tile * a.C + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(f >= b + 1)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { *a.errflag = 1; dead = true; }   // give up for good: results are invalid
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // flags observed before the partials are requested
          double tot[4] = {0.0, 0.0, 0.0, 0.0};
          for (int c0 = 0; c0 < a.C; c0 += 4) {                // 16 loads in flight per step
            double pv[4][4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
              const int c2 = (c0 + cc < a.C) ? c0 + cc : part;  // out-of-range and own slots read this part's own (valid) slot
#pragma unroll
              for (int r = 0; r < 4; r++)
                pv[cc][r] = __hip_atomic_load(&slot[(size_t)c2 * 256 + 64 * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (c0 + cc < a.C) tot[r] += (c0 + cc == part) ? own[r] : pv[cc][r];
          }
#pragma unroll
          for (int r = 0; r < 4; r++) Sp[lane + 64 * r] = tot[r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave reads it back below
        }
        if (NT <= 8) {
        // ---- sequential pass over the 16 SNPs (src/coreLoop.cpp:115-133).  lane = (g, col): col = trait, and the four
        // 16-lane groups g share everything outside the dependency chain: group g owns rows 4g .. 4g+3 of S (as in
        // aq_core_sweep_la.h; the in-block coupling uses the trait's own Gram block).  Rolled over the four owner groups
        // so that the residual tiles this wave also holds stay in registers. ----
        {
          const double *Gk = LGk + buf * (256 * 17) + col;
          double Sown[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int e = (4 * g + r) * 16 + col;
            double sv = Sp[e];
            if (a.C == 1) {
#pragma unroll
              for (int ww = 1; ww < NW; ww++) sv += Sp[ww * 256 + e];
            }
            Sown[r] = sv;
          }
          double sb = __shfl(Sown[0], col, 64);
          double m1o = Lm1[col], cA = LcA[col], cf = Lcoef[col], ci = Lci2s[col], dj = Lxn[col];
#pragma unroll 1
          for (int q4 = 0; q4 < 4; q4++) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
              const int j = 4 * q4 + jj;
              const int jn = (j + 1) & 15;
              const int en = jn * 16 + col;
              double m1o_n = Lm1[en], cA_n = LcA[en], cf_n = Lcoef[en], ci_n = Lci2s[en], d_n = Lxn[en];
              const double g_next = Gk[(jn * 16 + j) * 17];
              // the next SNP's S from its owner group, fetched ahead of the chain (owner q4, slot jj+1; or the next group's slot 0)
              const double s_next = __shfl(Sown[(jj + 1) & 3], (jn >> 2) * 16 + col, 64);
              double s = sb + m1o * dj;                         // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1 (cp_X - cp_X_rm_k)(j,j))   :121
              double mu = cf * s;                               // :125
              double x = fma(-(s * s), ci, cA);                 // :127-129 with mu^2 = coef^2 s^2 (keeps mu off the chain)
              double gm = aq_sigmoid_neg_fast(x);
              double dl = gm * mu - m1o;                        // m1 - m1_old   :130
              sb = s_next - g_next * dl;
              // in-block part of :132 with the trait's own Gram block, this group's rows (rows <= j are already consumed)
#pragma unroll
              for (int r = 0; r < 4; r++) Sown[r] -= Gk[((4 * g + r) * 16 + j) * 17] * dl;
              if (lane < 16) {
                Lgam[j * 16 + col] = gm;
                Lmu[j * 16 + col] = mu;
                Ldel[j * 16 + col] = dl;
              }
              m1o = m1o_n; cA = cA_n; cf = cf_n; ci = ci_n; dj = d_n;
            }
          }
        }
        } else {
          // (with 16 residual tiles per wave the four-group form spills; the plain lane = trait form is faster there)
        // ---- sequential pass over the 16 SNPs, lane = trait (src/coreLoop.cpp:115-133) ----
        if (lane < 16) {
          const double *Gk = LGk + buf * (256 * 17) + col;
          double S[16];
#pragma unroll
          for (int j = 0; j < 16; j++) {
            double s = Sp[j * 16 + col];
            if (a.C == 1) {
#pragma unroll
              for (int ww = 1; ww < NW; ww++) s += Sp[ww * 256 + j * 16 + col];
            }
            S[j] = s;
            if (j & 1) __builtin_amdgcn_sched_barrier(0);   // bound the LDS loads in flight (VGPR budget)
          }
          double m1o = Lm1[col], cA = LcA[col], cf = Lcoef[col], ci = Lci2s[col], dj = Lxn[col];
#pragma unroll 1
          for (int j = 0; j < 16; j++) {
            const int jn = ((j + 1) & 15) * 16 + col;
            double m1o_n = Lm1[jn], cA_n = LcA[jn], cf_n = Lcoef[jn], ci_n = Lci2s[jn], d_n = Lxn[jn];
            double s = S[0] + m1o * dj;                       // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1 (cp_X - cp_X_rm_k)(j,j))   :121
            double mu = cf * s;                               // :125
            double x = fma(-(s * s), ci, cA);                 // :127-129 with mu^2 = coef^2 s^2 (keeps mu off the chain)
            double gm = aq_sigmoid_neg_fast(x);
            double dl = gm * mu - m1o;                        // m1 - m1_old   :130
            // in-block part of :132 with the trait's own Gram block; rows past the block are never consumed
#pragma unroll
            for (int i0 = 0; i0 < 15; i0 += 8) {
              double gc[8];
#pragma unroll
              for (int i = 0; i < 8; i++) {
                int row = j + 1 + i0 + i;
                row = row < 15 ? row : 15;
                gc[i] = (i0 + i < 15) ? Gk[(row * 16 + j) * 17] : 0.0;
              }
#pragma unroll
              for (int i = 0; i < 8; i++)
                if (i0 + i < 15) S[i0 + i] = S[i0 + i + 1] - gc[i] * dl;
              __builtin_amdgcn_sched_barrier(0);
            }
            Lgam[j * 16 + col] = gm;
            Lmu[j * 16 + col] = mu;
            Ldel[j * 16 + col] = dl;
            m1o = m1o_n; cA = cA_n; cf = cf_n; ci = ci_n; dj = d_n;
          }
        }
        }
      } else if (more) {
        compute_gk(b + 1, buf ^ 1);
      }
      aq_lds_barrier();

      // ---- finalize block b (helper threads): stores and column/row sums ----
      if (helper) {
        const int j = 16 * b + hj;
        double gm = Lgam[tid], mu = Lmu[tid];
        size_t off = tbase + (size_t)(16 * b) * 16 + tid;
        double gbv = 0.0;
        if (lead) {
          a.gam[off] = gm;
          a.mu[off] = mu;
        }
        if (kvalid && j < a.p) {
          double be = gm * mu;
          double m2 = (mu * mu + Ls2[tid]) * gm;              // update_m2_beta_, R/update_vb.R:19-31
          gbv = gm * LB[tid];
          Lred[tid] += gm;
          Lred[256 + tid] += m2;
          Lred[512 + tid] += Lxn[tid] * (m2 - be * be);       // kappa's X_norm_sq terms, R/update_vb.R:152-154
          Lred[768 + tid] += gbv;
          Lred[1024 + tid] += gm * Lls2[tid];
        }
        gbv = aq_row16_sum(gbv);
        if (hk == 0 && lead) a.rowGB[(size_t)tile * a.p_pad + j] = gbv;
      }

      // ---- R -= mis .* (X_b delta), and S of block b+1, tile by tile ----
      double nd[4];
#pragma unroll
      for (int s = 0; s < 4; s++) nd[s] = -Ldel[(4 * s + g) * 16 + col];
      acc = (aq_d4){0, 0, 0, 0};
      const double2 *xu = XUw + (size_t)b * NTT * 128;
      const double2 *xa = XAw + (size_t)(more ? b + 1 : b) * NTT * 128;
      double2 cu0 = xu[0], cu1 = xu[64], ca0 = xa[0], ca1 = xa[64];
      aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        double2 nu0, nu1, na0, na1;
        if (t + 1 < NT) {
          nu0 = xu[(t + 1) * 128]; nu1 = xu[(t + 1) * 128 + 64];
          na0 = xa[(t + 1) * 128]; na1 = xa[(t + 1) * 128 + 64];
        }
        Rr[t] = aq_mfma(cu0.x, nd[0], Rr[t]);
        Rr[t] = aq_mfma(cu0.y, nd[1], Rr[t]);
        Rr[t] = aq_mfma(cu1.x, nd[2], Rr[t]);
        Rr[t] = aq_mfma(cu1.y, nd[3], Rr[t]);
        remask(tc);
        acc = aq_mfma(ca0.x, Rr[t][0], acc);
        acc = aq_mfma(ca0.y, Rr[t][1], acc);
        acc = aq_mfma(ca1.x, Rr[t][2], acc);
        acc = aq_mfma(ca1.y, Rr[t][3], acc);
        if (t + 1 < NT) { cu0 = nu0; cu1 = nu1; ca0 = na0; ca1 = na1; }
        asm volatile("" ::: "memory");   // the next tile's operand loads stay behind this one's (no hoisting of the whole stream)
        __builtin_amdgcn_sched_barrier(0);
      });
      if (helper && more) stage_commit(buf ^ 1);
    }
  }

  // ---- residual back, ||R_k||^2, column sums ----
  __syncthreads();
  {
    // (lane id and base pointer re-derived behind an optimisation barrier: otherwise the load addresses of the prologue are
    // kept alive -- spilled -- across the whole sweep)
    unsigned zero2 = 0;
    asm volatile("" : "+v"(zero2));
    const int ln2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero2));
    double *Rout = a.R;
    asm volatile("" : "+s"(Rout));
    double *Rg = Rout + (size_t)tile * a.n_pad * 16 + (size_t)(16 * wt0 + (ln2 >> 4)) * 16 + (ln2 & 15);
    double rn = 0.0;
    aq_static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double v = Rr[t][r];
        Rg[(16 * t + 4 * r) * 16] = v;
        rn += v * v;
      }
    });
    Lrn[(w * 4 + (ln2 >> 4)) * 16 + (ln2 & 15)] = rn;
  }
  __syncthreads();
  if (tid < 16) {
    const int kk = tile * 16 + tid;
    const size_t Q = a.q_pad;
    double r2 = 0.0;
#pragma unroll 1
    for (int s = 0; s < NW * 4; s++) r2 += Lrn[s * 16 + tid];
    double *sm = a.sums + (size_t)seg * 6 * Q;             // per-segment slot (combined by aq_k_combine_segment_sums6)
    if (a.C > 1) a.rnpart[(size_t)part * Q + kk] = r2;   // added over the parts by aq_k_sum_parts
    else sm[4 * Q + kk] = r2;
#pragma unroll 1
    for (int u = 0; u < 5 && lead; u++) {   // Lred rows: gam, m2, sx, gam*b, gam*log sig2_beta -> sums rows 0,1,2,3,5
      double acc2 = 0.0;
#pragma unroll 1
      for (int jj = 0; jj < 16; jj++) acc2 += Lred[u * 256 + jj * 16 + tid];
      sm[(size_t)(u < 4 ? u : 5) * Q + kk] = acc2;
    }
  }
  if (a.nseg > 1) {
    // publish this tile's residual for the next segment: stores drained, one agent-scope release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.done[tile], seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// NA-form sums (6 rows per slot): rows 0-3 and 5 added over the chained segments, ||R||^2 (row 4) from the last one
__global__ void aq_k_combine_segment_sums6(double *sums, int q_pad, int nslot) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= q_pad) return;
  size_t Q = q_pad;
  for (int v = 0; v < 6; v++) {
    if (v == 4) continue;
    double acc = sums[v * Q + k];
    for (int s = 1; s < nslot; s++) acc += sums[(size_t)s * 6 * Q + v * Q + k];
    sums[v * Q + k] = acc;
  }
  sums[4 * Q + k] = sums[(size_t)(nslot - 1) * 6 * Q + 4 * Q + k];
}

// X_b in row-major panels for the gathers: XR[(b*NR + i)*16 + jj] = x_{i, 16 b + jj} (0 beyond n or p)
__global__ void aq_k_build_xr(const double *__restrict__ X, double *__restrict__ XR, int n, int p, int nb, int NR) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nb * NR * 16;
  if (e >= total) return;
  int jj = (int)(e & 15);
  size_t rest = e >> 4;
  int i = (int)(rest % NR);
  int b = (int)(rest / NR);
  int j = 16 * b + jj;
  XR[e] = (i < n && j < p) ? X[(size_t)i + (size_t)n * j] : 0.0;
}

// sums[4][k] = sum over the C sample parts of their ||R_k||^2 (fixed order)
__global__ void aq_k_sum_parts(const double *__restrict__ rnpart, double *__restrict__ dst, int C, int q_pad) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= q_pad) return;
  double s = 0.0;
  for (int c = 0; c < C; c++) s += rnpart[(size_t)c * q_pad + k];
  dst[k] = s;
}

// ---- per-trait Gram blocks for the masked form of the look-ahead kernel (AqCoreArgs::GK) -------------------------------------
// For every trait tile and SNP block b, and each of the tile's 16 traits k:
//     diagonal block  X_b' diag(mis_k) X_b     = X_b'X_b     - Xm_k(b)' Xm_k(b)        lower triangle [i (i + 1) / 2 + j][k]
//     cross block     X_b' diag(mis_k) X_{b-1} = X_b'X_{b-1} - Xm_k(b)' Xm_k(b-1)      [i][j][k]
// (Xm_k(b) = the rows of X_b at trait k's missing samples: rank-m_k f64-MFMA corrections from gathered 128-byte row segments,
// as compute_gk above).  They depend on X and on the missingness pattern only, so they are computed ONCE per handle and kept in
// HBM -- 50 KB per (tile, block), 98 GB for a C5 trait shard: what 288 GB are for -- instead of being recomputed by every sweep
// (2 m_k 256 flop per trait and block, a third of the sweep's MFMA work at 5 % missing, and the cross blocks would double it).
// grid = (ceil(nb / bchunk), ntile), 512 threads; wave w handles the jobs (trait, kind) = w, w + 8, ... of each block.
__global__ __launch_bounds__(512) void aq_k_gk_blocks(const double *__restrict__ XR, const double *__restrict__ G,
                                                    const double *__restrict__ Gx, const int *__restrict__ midx,
                                                    const int *__restrict__ mcnt4, double *__restrict__ GK, int nb, int NR, int Mmax,
                                                    int bchunk) {
  extern __shared__ unsigned short aq_gk_lidx[];   // [16][Mmax]
  __shared__ int Lcnt[16];
  const int tile = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, col = lane & 15;
  for (int e = tid; e < 16 * Mmax; e += 512) aq_gk_lidx[e] = (unsigned short)midx[(size_t)tile * 16 * Mmax + e];
  if (tid < 16) Lcnt[tid] = mcnt4[tile * 16 + tid];
  __syncthreads();
  const int b0 = blockIdx.x * bchunk, b1 = min(nb, b0 + bchunk);
  for (int b = b0; b < b1; b++) {
    double *out = GK + ((size_t)tile * nb + b) * AQ_GK_STRIDE;
    const double *xr = XR + (size_t)b * NR * 16 + col;
    const double *xp = XR + (size_t)(b > 0 ? b - 1 : 0) * NR * 16 + col;
    for (int job = w; job < 32; job += 8) {
      const int k = job & 15, cross = job >> 4;
      aq_d4 acc = (aq_d4){0, 0, 0, 0};
      const int n4 = (cross && b == 0) ? 0 : Lcnt[k];
      const unsigned short *ix = aq_gk_lidx + k * Mmax + g;
      for (int t = 0; t < n4; t += 4) {   // lists are padded to whole groups of 16 samples (index n_pad = an all-zero row)
        const int i0 = ix[4 * t], i1 = ix[4 * t + 4], i2 = ix[4 * t + 8], i3 = ix[4 * t + 12];
        const double a0 = xr[(size_t)i0 * 16], a1 = xr[(size_t)i1 * 16], a2 = xr[(size_t)i2 * 16], a3 = xr[(size_t)i3 * 16];
        if (cross) {
          const double c0 = xp[(size_t)i0 * 16], c1 = xp[(size_t)i1 * 16], c2 = xp[(size_t)i2 * 16], c3 = xp[(size_t)i3 * 16];
          acc = aq_mfma(a0, c0, acc); acc = aq_mfma(a1, c1, acc); acc = aq_mfma(a2, c2, acc); acc = aq_mfma(a3, c3, acc);
        } else {
          acc = aq_mfma(a0, a0, acc); acc = aq_mfma(a1, a1, acc); acc = aq_mfma(a2, a2, acc); acc = aq_mfma(a3, a3, acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = 4 * r + g, j = col;            // D layout: row = 4 reg + (lane >> 4), column = lane & 15
        if (cross) {
          const double base = b > 0 ? Gx[(size_t)b * 256 + i * 16 + j] : 0.0;
          out[AQ_GK_DIAG + (i * 16 + j) * 16 + k] = base - acc[r];
        } else if (i >= j) {
          out[(i * (i + 1) / 2 + j) * 16 + k] = G[(size_t)b * 256 + i * 16 + j] - acc[r];
        }
      }
    }
  }
}
