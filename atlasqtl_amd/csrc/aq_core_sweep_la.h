// aq_core_sweep_la.h -- core sweep kernel, "look-ahead" form (the one bench.py measures).
//
// Same arithmetic as aq_core_sweep.h (blocked Gauss-Seidel in n-space for
// src/coreLoop.cpp:38-86), restructured so that the sequential 16-SNP pass -- a chain of
// ~23 dependent fp64 operations per SNP that cannot be shortened -- runs CONCURRENTLY with the
// matrix work instead of between two barriers:
//
//   workgroup = 8 waves: 6 "matrix" waves (each owns exactly NT 16-sample tiles of the residual R_K in VGPRs)
//             + 1 "recurrence" wave (lane = trait, owns no residual) + 1 "helper" wave (the block's loads, stores and
//             column sums).  Waves i and i+4 of a workgroup share a SIMD: the recurrence wave is wave 3 and the helper
//             wave 7, so no f64 MFMA is ever issued on the recurrence wave's SIMD -- its chain of dependent fp64
//             operations would otherwise wait behind every one of them (measured: 2.6x slower) -- and the six matrix
//             waves (0,1,2,4,5,6) run a pure MFMA stream on the other three SIMDs
//   phase b (one barrier per phase):
//     recurrence wave : SNP block b.   s_j = S'_b[j] - (X_b'X_{b-1} delta_{b-1})[j]      cross-block Gram, precomputed
//                                            - sum_{i<j} (X_b'X_b)[j,i] delta_i            in-block Gram
//                       then mu, gam, m1, delta_j as src/coreLoop.cpp:69-79
//     matrix waves    : R_K -= X_{b-1} delta_{b-1}   (update of the block finished one phase ago)
//                       S'_{b+1} = X_{b+1}' R_K        (f64 MFMA, k = samples)
//                       + the block's loads/stores and column sums on their otherwise idle VALU/LSU
//   S'_{b+1} misses only the update of block b, which the recurrence wave adds as the 16x16
//   cross-Gram correction, so the result is the same Gauss-Seidel sweep.
//
// LDS buffers are double-buffered by block parity; every hand-off crosses exactly one barrier.
#pragma once
#include <hip/hip_runtime.h>
#include "aq_core_sweep.h"
#include <type_traits>

// NT = residual tiles of matrix waves 0,1,2; NT2 (= NT or NT-1) those of waves 4,5,6: each SIMD carries NT + NT2.
// SEG: chained-segment launch (a.nseg * a.ntile workgroups).  Workgroup s*ntile + k handles SNP segment s of trait
// tile k, starting from the residual that segment s-1 of the same tile left in global memory.  Blocks are dispatched
// in index order, so that workgroup has normally finished long before; correctness does not depend on it: the
// hand-off is an agent-scope release (producer) / acquire (consumer) around done[k], and the wait is bounded.
// With one workgroup per CU this turns 625 tiles on 256 CUs from 3 whole rounds into 2.44 + rounding.
template <int NT, int NT2, bool SEG>
__global__ __launch_bounds__(8 * 64, 2) void aq_core_sweep_la_kernel(const AqCoreArgs a) {
  constexpr int NWM = 6;                        // matrix waves: 0,1,2,4,5,6
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const bool is_rec = (w == 3);
  const int mw = w < 3 ? w : w - 1;             // matrix-wave index 0..5
  const int g = lane >> 4;
  const int col = lane & 15;
  int tile_ = a.tile_first + blockIdx.x, seg_b0 = a.b_begin, seg_b1 = a.b_end, seg_slot = a.sums_slot, seg = 0;
  if (SEG) {
    seg = blockIdx.x / a.ntile;
    tile_ = blockIdx.x - seg * a.ntile;
    seg_b0 = (int)((long long)a.nb * seg / a.nseg);
    seg_b1 = (int)((long long)a.nb * (seg + 1) / a.nseg);
    seg_slot = seg;
    if (seg > 0) {
      if (tid == 0) {
        int tries = 0;
        while (__hip_atomic_load(&a.done[tile_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg) {
          __builtin_amdgcn_s_sleep(32);
          if (++tries > 4000000) { *a.errflag = 1; break; }   // bounded: never hang the GPU
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
  }
  const int tile = tile_;
  const bool helper = (w == 7);                 // helper wave: entry e = lane + 64 r (r = 0..3) <-> (snp e >> 4, trait e & 15)
  const int hk = lane & 15, hj0 = lane >> 4;
  const int mr = a.dmode ? 1 : 4, mg = a.dmode ? 4 : 1;   // f64 MFMA D row = mr*reg + mg*(lane>>4)

  __shared__ double Sp[2][NWM][256];   // partial S' of each matrix wave [snp][trait]
  __shared__ double LA[2][256];        // A = log(1-Phi) - log Phi
  __shared__ double Lm1[2][256];       // old m1 = gam*mu
  __shared__ double LB[2][256];        // slope b of Z
  __shared__ double Laa[2][256];       // intercept a of Z (Z = a + gam b, R/update_vb.R:217-234)
  __shared__ double LG[2][512];        // X_b'X_b as [16][32], upper 16 columns zero
  __shared__ double LGx[2][256];       // X_b'X_{b-1}  [j][i]
  __shared__ double Lgam[2][256], Lmu[2][256], Ldel[2][256];
  __shared__ double Lred[4][256];      // running column sums per helper thread
  __shared__ double Lrn[NWM * 4][16];
  // Point-to-point progress counters instead of a workgroup barrier per phase (sweep mode): Fl[0..5] = number of SNP
  // blocks whose partial S' matrix wave m has written, Fl[6] = blocks the recurrence wave has finished, Fl[7] = blocks
  // the helper wave has staged.  Each wave waits only for what it really reads, so the matrix waves -- the critical
  // path -- never stop at a barrier (it cost them 0.3 us of LDS congestion after, 0.3 us of MFMA drain before and the
  // skew of eight waves, per 5.9 us phase).  LDS operations of a wave execute in order and the LDS is one pipeline per
  // CU, so "data stores; s_waitcnt lgkmcnt(0); counter store" on one side and "counter load ... ; data loads" on the
  // other are ordered; the asm memory clobbers keep the compiler from moving accesses across them.
  __shared__ int Fl[8];
  // (explicit LDS address space: through a generic pointer the volatile accesses become flat loads with a vmcnt(0) drain)
  typedef __attribute__((address_space(3))) volatile int aq_lds_vint;
  aq_lds_vint *Flv = (aq_lds_vint *)(__attribute__((address_space(3))) int *)Fl;
  auto signal = [&](int idx, int val) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) Flv[idx] = val;
  };
  auto wait_ge = [&](int idx, int val) {
    while (Flv[idx] < val) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
  };
  if (tid < 8) Fl[tid] = 0;

  const size_t tbase = (size_t)tile * a.p_pad * 16;
  double *Rg = a.R + (size_t)tile * a.n_pad * 16;
  const int ktrait = tile * 16 + hk;
  const bool kvalid = ktrait < a.q;
  double sig2b_k = 1.0;
  if (helper) {
    sig2b_k = a.sig2b[ktrait];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int e = lane + 64 * r, hj = hj0 + 4 * r;
      Lred[0][e] = Lred[1][e] = Lred[2][e] = Lred[3][e] = 0.0;
      LG[0][hj * 32 + 16 + hk] = 0.0;
      LG[1][hj * 32 + 16 + hk] = 0.0;
    }
  }

  __syncthreads();   // counters and the helper's LDS initialisation are visible to every role
  if (is_rec) {
    // =========================== recurrence wave ===========================================

    if (a.mode == 1) {
      for (int b = seg_b0; b < seg_b1; b++) {   // init mode: nothing to do, keep the barrier count
        __syncthreads();
        __syncthreads();
      }
    } else {
      const int kk = tile * 16 + col;
      const double rc_coef = a.coef[kk];
      const double rc_cinv2s = a.c * a.inv2s[kk];
      const double rc_cst = a.cst[kk];
      const double rc_K = rc_coef * rc_coef * rc_cinv2s;   // keeps mu off the dependency chain of the recursion
      for (int b = seg_b0; b < seg_b1; b++) {
        const int par = b & 1;
        {   // block b needs its six partial S' and its staged scalars
          const int need = b - seg_b0 + 1;
#pragma unroll
          for (int m = 0; m < NWM; m++) wait_ge(m, need);
          wait_ge(7, need);
        }
#if AQ_DIAG & 8
        const long long t_in = clock64();
#endif
#if !(AQ_DIAG & 1)
        // ---- SNP block b.  lane = (g, col): col = trait, g = lane >> 4 one of four row groups ----------------------
        // This wave and the helper wave share one SIMD and both are bound by its fp64 VALU issue rate (a dependent
        // v_fma_f64 has a latency of only 6 cycles, tools/microbench/f64_valu.hip), so everything that does not belong
        // to the chain itself is spread over the four 16-lane groups instead of being repeated in each of them:
        // group g owns the rows g, g+4, g+8, g+12 of S.
        double Sown[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int j = g + 4 * r;
          double sv = Sp[par][0][j * 16 + col];
#pragma unroll
          for (int ww = 1; ww < NWM; ww++) sv += Sp[par][ww][j * 16 + col];
          Sown[r] = sv;
        }
        if (b > seg_b0) {
          // cross-block correction X_b'X_{b-1} delta_{b-1} of this group's rows (a segment starts from a complete residual)
          double dlp[16];
#pragma unroll
          for (int i = 0; i < 16; i++) dlp[i] = Ldel[par ^ 1][i * 16 + col];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double *gx = &LGx[par][(g + 4 * r) * 16];
            double cx = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) cx += gx[i] * dlp[i];
            Sown[r] -= cx;
          }
        }
        double sb = __shfl(Sown[0], col, 64);     // S of SNP 0 (group 0) to every group
        double m1o = Lm1[par][col], cA = a.c * (LA[par][col] + rc_cst), dj = LG[par][0];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int jn = (j + 1) & 15;
          double m1o_n = Lm1[par][jn * 16 + col], cA_n = a.c * (LA[par][jn * 16 + col] + rc_cst), d_n = LG[par][jn * 33];
          // the next SNP's S is fetched from its owner group BEFORE this step's delta is known (off the chain) ...
          const double g_next = LG[par][j * 32 + jn];
          const double s_next = __shfl(Sown[jn >> 2], (jn & 3) * 16 + col, 64);
          double s = sb + m1o * dj;                         // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1*cp_X(j,j))   :71
          double mu = rc_coef * s;                          // :73
          double x = fma(-(s * s), rc_K, cA);               // c*(log(1-Phi) - log Phi - mu^2/(2 sig2) + cst), mu^2 = coef^2 s^2   :75-77
          double gm = aq_sigmoid_neg_fast(x);
          double dl = gm * mu - m1o;                        // m1 - m1_old, m1 = gam*mu   :79
          sb = s_next - g_next * dl;                        // ... and completed with this step's update: one FMA on the chain
          // in-block part of :81, this group's rows (rows <= j are already consumed: updating them is harmless)
#pragma unroll
          for (int r = (j >> 2); r < 4; r++) Sown[r] -= LG[par][j * 32 + g + 4 * r] * dl;
          if (lane < 16) {
            Lgam[par][j * 16 + col] = gm;
            Lmu[par][j * 16 + col] = mu;
            Ldel[par][j * 16 + col] = dl;
          }
          m1o = m1o_n; cA = cA_n; dj = d_n;
        }
#endif
#if AQ_DIAG & 8
        const long long t_work = clock64();
#endif
        signal(6, b - seg_b0 + 1);   // delta, gam, mu of block b are in LDS
#if AQ_DIAG & 8
        if (blockIdx.x == 0 && lane == 0 && a.dbg && (b - seg_b0) < 256) {
          long long *d = a.dbg + ((size_t)(b - seg_b0) * 16 + w) * 3;
          d[0] = t_in; d[1] = t_work; d[2] = clock64();
        }
#endif
      }
    }
    __syncthreads();   // matches the matrix waves' barrier before the final sums
  } else if (helper) {
    // =========================== helper wave ===============================================
    // global -> registers a phase ahead -> LDS; stores and column / row sums of finished blocks
    // The transcendental per-entry inputs of the block are computed HERE, a phase ahead, from theta_j + zeta_k (the work
    // of the former p x q pre-pass kernel, which wrote and re-read 16 B per entry):
    //   A = log(1-Phi(u)) - log Phi(u)      src/coreLoop.cpp:75-76 (its log_Phi / log_1_min_Phi inputs, R/...core.R:293-295)
    //   Z = a + gam b,  a = u + imr0/sqrt(c), b = (imr1 - imr0)/sqrt(c) at U = sqrt(c) u       R/update_vb.R:217-234
    // This wave shares its SIMD only with the recurrence wave, whose dependent chain leaves the VALU mostly idle.
    double st_A[4], st_g[4], st_m[4], st_B[4], st_a[4], st_G[4], st_Gx[4];
    const double zk = a.zeta[ktrait];
    double th[4];   // theta of the block to be staged next, loaded a phase earlier so that the arithmetic never waits for HBM
    auto theta_load = [&](int b) {
#pragma unroll
      for (int r = 0; r < 4; r++) th[r] = a.theta[16 * b + hj0 + 4 * r];
    };
    auto stage_load = [&](int b) {   // memory part: issue the loads, no waiting
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = lane + 64 * r;
        size_t off = tbase + (size_t)(16 * b) * 16 + e;
        st_g[r] = a.gam[off];
        st_m[r] = a.mu[off];
        st_G[r] = a.G[(size_t)b * 256 + e];
        st_Gx[r] = a.Gx[(size_t)b * 256 + e];
      }
    };
    auto stage_probit = [&](int b) {   // arithmetic part, from th[] (block b)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int j = 16 * b + hj0 + 4 * r;
        const double u = th[r] + zk;
        double A, imr1, imr0, ee;
#if AQ_DIAG & 4
        A = u; imr1 = 1.0; imr0 = -1.0; ee = 0.0;        // timing diagnostics: no transcendental work in the helper wave
#else
        aq_probit_A_imr(u, &A, &imr1, &imr0, &ee);
#endif
        if (!a.c_is_one) {
          double Ac;
          aq_probit_A_imr(a.sqrt_c * u, &Ac, &imr1, &imr0, &ee);
          imr1 /= a.sqrt_c;
          imr0 /= a.sqrt_c;
        }
        const bool valid = kvalid && j < a.p;
        st_A[r] = valid ? A : 0.0;
        st_B[r] = valid ? imr1 - imr0 : 0.0;
        st_a[r] = valid ? u + imr0 : 0.0;
      }
    };
    auto stage_commit = [&](int par) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = lane + 64 * r, hj = hj0 + 4 * r;
        LA[par][e] = st_A[r];
        Lm1[par][e] = st_g[r] * st_m[r];
        LB[par][e] = st_B[r];
        Laa[par][e] = st_a[r];
        LG[par][hj * 32 + hk] = st_G[r];
        LGx[par][e] = st_Gx[r];
      }
    };
    auto finalize = [&](int b, int par) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = lane + 64 * r, hj = hj0 + 4 * r;
        double gm = Lgam[par][e], mu = Lmu[par][e];
        size_t off = tbase + (size_t)(16 * b) * 16 + e;
        a.gam[off] = gm;
        a.mu[off] = mu;
        const int j = 16 * b + hj;
        double gb = 0.0;
        if (kvalid && j < a.p) {
          double be = gm * mu;
          gb = Laa[par][e] + gm * LB[par][e];       // Z_jk = a + gam b: its row and column sums are all that is needed
          Lred[0][e] += gm;
          Lred[1][e] += (mu * mu + sig2b_k) * gm;   // update_m2_beta_, R/update_vb.R:19-31
          Lred[2][e] += be * be;
          Lred[3][e] += gb;
        }
        gb += __shfl_xor(gb, 8, 64);
        gb += __shfl_xor(gb, 4, 64);
        gb += __shfl_xor(gb, 2, 64);
        gb += __shfl_xor(gb, 1, 64);
        if (hk == 0) a.rowGB[(size_t)tile * a.p_pad + j] = gb;
      }
    };
    if (a.mode == 1) {
      for (int b = seg_b0; b < seg_b1; b++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int e = lane + 64 * r, hj = hj0 + 4 * r;
          size_t off = tbase + (size_t)(16 * b) * 16 + e;
          double gm = a.gam[off], mu = a.mu[off];
          double be = gm * mu;                                  // update_beta_vb_, R/update_vb.R:17
          Ldel[0][e] = be;
          if (kvalid && (16 * b + hj) < a.p) {
            Lred[0][e] += gm;
            Lred[1][e] += (mu * mu + sig2b_k) * gm;             // initial m2_beta, R/atlasqtl_global_local_core.R:113
            Lred[2][e] += be * be;
          }
        }
        __syncthreads();
        __syncthreads();
      }
    } else {
      theta_load(seg_b0);
      stage_load(seg_b0);
      stage_probit(seg_b0);
      if (seg_b0 + 1 < seg_b1) theta_load(seg_b0 + 1);
      stage_commit(seg_b0 & 1);
      signal(7, 1);
      for (int b = seg_b0; b < seg_b1; b++) {
        const int par = b & 1;
        const bool more = (b + 1 < seg_b1);
#if AQ_DIAG & 8
        const long long t_in = clock64();
#endif
        if (more) stage_load(b + 1);
        if (more) stage_probit(b + 1);
        if (b + 2 < seg_b1) theta_load(b + 2);
        // block b-1 must be through the recurrence: its gam / mu are read here, and the parity buffers about to be
        // overwritten with block b+1 are the ones it read
        if (b > seg_b0) { wait_ge(6, b - seg_b0); finalize(b - 1, par ^ 1); }
        if (more) { stage_commit(par ^ 1); signal(7, b - seg_b0 + 2); }
#if AQ_DIAG & 8
        const long long t_work = clock64();
#endif
#if AQ_DIAG & 8
        if (blockIdx.x == 0 && lane == 0 && a.dbg && (b - seg_b0) < 256) {
          long long *d = a.dbg + ((size_t)(b - seg_b0) * 16 + w) * 3;
          d[0] = t_in; d[1] = t_work; d[2] = clock64();
        }
#endif
      }
      wait_ge(6, seg_b1 - seg_b0);
      finalize(seg_b1 - 1, (seg_b1 - 1) & 1);
    }
    __syncthreads();   // matches the matrix waves' barrier before the final sums
  } else {
    // =========================== matrix waves ==============================================
    // residual tiles: Rr[t][r] <-> sample 16*(my_t0+t) + mr*r + mg*g, trait col
    constexpr int NTT = 3 * (NT + NT2);
    const bool hi = mw < 3;                          // owns NT tiles (else NT2)
    const int my_t0 = hi ? mw * NT : 3 * NT + (mw - 3) * NT2;
    aq_d4 Rr[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        Rr[t][r] = (hi || t < NT2) ? Rg[(size_t)(16 * (my_t0 + t) + mr * r + mg * g) * 16 + col] : 0.0;
    const double2 *XAw = a.XA + (size_t)my_t0 * 128 + lane;
    const double2 *XUw = a.XU + (size_t)my_t0 * 128 + lane;

    // matrix work of one phase: update with block bu (delta in LDS parity pu) and/or S' of block bs -> Sp[ps]
    // operands of residual tile 0 of the NEXT phase, requested before the barrier so that no phase starts with an exposed load
    double2 fu0, fu1, fa0, fa1;
    auto prefetch_first = [&](int bu, bool do_s, int bs) {
      const double2 *xu = XUw + (size_t)bu * NTT * 128;
      const double2 *xa = XAw + (size_t)(do_s ? bs : 0) * NTT * 128;
      fu0 = xu[0]; fu1 = xu[64]; fa0 = xa[0]; fa1 = xa[64];
    };
    auto matrix_phase_n = [&](auto ntc, bool do_u, int bu, int pu, bool do_s, int bs, int ps, bool pre) {
      constexpr int NTC = decltype(ntc)::value;
      double nd[4];
      if (do_u) {
        // (4 s + g) 16 + col = 64 s + lane.  The lane id is recomputed here and hidden from the optimiser: kept in a
        // register across the phase it was spilled, and the reload's s_waitcnt vmcnt(0) drained the operand prefetch of
        // every phase -- 0.5 us of exposed L2 latency.
        unsigned zero = 0;
        asm volatile("" : "+v"(zero));   // opaque input: the two v_mbcnt are re-issued every phase instead of being hoisted and spilled
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const double *dl = &Ldel[pu][ln];
#pragma unroll
        for (int s = 0; s < 4; s++) nd[s] = -dl[64 * s];
      }
      aq_d4 acc = {0, 0, 0, 0};
      const double2 *xu = XUw + (size_t)(do_u ? bu : 0) * NTT * 128;
      const double2 *xa = XAw + (size_t)(do_s ? bs : 0) * NTT * 128;
      double2 cu0, cu1, ca0, ca1;
      if (pre) { cu0 = fu0; cu1 = fu1; ca0 = fa0; ca1 = fa1; }   // tile 0 was requested before the previous barrier
      else { cu0 = xu[0]; cu1 = xu[64]; ca0 = xa[0]; ca1 = xa[64]; }
      // S' runs one tile behind the update: step t issues U(t) (4 MFMAs chained on Rr[t]) and then S'(t-1) (4 chained on
      // acc), whose B operand Rr[t-1] was finished a whole step earlier -- a wave that runs alone on its SIMD (the two waves
      // of a SIMD are ~1.3 us apart) no longer waits one MFMA latency per tile.  Same registers, same prefetch distance:
      // XA(t) is requested one step later than before, together with XU(t+1).
#pragma unroll
      for (int t = 0; t < NTC; t++) {
        double2 nu0, nu1, na0, na1;
        if (t + 1 < NTC) { nu0 = xu[(t + 1) * 128]; nu1 = xu[(t + 1) * 128 + 64]; }
        if (t >= 1) { na0 = xa[t * 128]; na1 = xa[t * 128 + 64]; }
        if (do_u) {
          aq_d4 Rt = Rr[t];
          Rt = aq_mfma(cu0.x, nd[0], Rt);
          Rt = aq_mfma(cu0.y, nd[1], Rt);
          Rt = aq_mfma(cu1.x, nd[2], Rt);
          Rt = aq_mfma(cu1.y, nd[3], Rt);
          Rr[t] = Rt;
        }
        if (t >= 1) {
          if (do_s) {
            const aq_d4 Rp = Rr[t - 1];
            acc = aq_mfma(ca0.x, Rp[0], acc);
            acc = aq_mfma(ca0.y, Rp[1], acc);
            acc = aq_mfma(ca1.x, Rp[2], acc);
            acc = aq_mfma(ca1.y, Rp[3], acc);
          }
          ca0 = na0; ca1 = na1;
        }
        if (t + 1 < NTC) { cu0 = nu0; cu1 = nu1; }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (do_s) {
        const aq_d4 Rp = Rr[NTC - 1];
        acc = aq_mfma(ca0.x, Rp[0], acc);
        acc = aq_mfma(ca0.y, Rp[1], acc);
        acc = aq_mfma(ca1.x, Rp[2], acc);
        acc = aq_mfma(ca1.y, Rp[3], acc);
      }
      if (do_s) {
#pragma unroll
        for (int i = 0; i < 4; i++) Sp[ps][mw][(mr * i + mg * g) * 16 + col] = acc[i];
      }
    };
    auto matrix_phase = [&](bool do_u, int bu, int pu, bool do_s, int bs, int ps, bool pre) {
      if (NT2 == NT || hi) matrix_phase_n(std::integral_constant<int, NT>{}, do_u, bu, pu, do_s, bs, ps, pre);
      else matrix_phase_n(std::integral_constant<int, NT2>{}, do_u, bu, pu, do_s, bs, ps, pre);
    };

    if (a.mode == 1) {
      // ---------------- init mode: R = Y - X (gam*mu), block by block; column sums of the initial state
      for (int b = seg_b0; b < seg_b1; b++) {
        __syncthreads();                                  // the helper wave has put beta of block b into Ldel[0]
        matrix_phase(true, b, 0, false, 0, 0, false);
        __syncthreads();
      }
    } else {
      // ---------------- full sweep -----------------------------------------------------
      // prologue: S'_0 from the untouched residual, staging of block 0
      matrix_phase(false, 0, 0, true, seg_b0, seg_b0 & 1, false);
      signal(mw, 1);
      prefetch_first(0, seg_b0 + 1 < seg_b1, seg_b0 + 1);   // first phase: no update yet, S' of block seg_b0 + 1
      for (int b = seg_b0; b < seg_b1; b++) {
        const int par = b & 1;
        const bool more = (b + 1 < seg_b1);
        // update with block b-1 (its delta must be out: recurrence counter), S' of block b+1
#if AQ_DIAG & 8
        const long long t_in = clock64();
#endif
        if (b > seg_b0) wait_ge(6, b - seg_b0);
#if !(AQ_DIAG & 2)
        if (b > seg_b0 || more) matrix_phase(b > seg_b0, b - 1, par ^ 1, more, b + 1, par ^ 1, true);
        if (more) signal(mw, b - seg_b0 + 2);
        prefetch_first(b, b + 2 < seg_b1, b + 2);   // next phase (or the epilogue): update with block b, S' of block b + 2
#else
        if (more) signal(mw, b - seg_b0 + 2);
#endif
#if AQ_DIAG & 8
        const long long t_work = clock64();
#endif
#if AQ_DIAG & 8
        if (blockIdx.x == 0 && lane == 0 && a.dbg && (b - seg_b0) < 256) {
          long long *d = a.dbg + ((size_t)(b - seg_b0) * 16 + w) * 3;
          d[0] = t_in; d[1] = t_work; d[2] = clock64();
        }
#endif
      }
      // epilogue: the last block's update and stores
      const int pl = (seg_b1 - 1) & 1;
      wait_ge(6, seg_b1 - seg_b0);
#if !(AQ_DIAG & 2)
      matrix_phase(true, seg_b1 - 1, pl, false, 0, 0, true);
#endif
    }
    // ---- write the residual back and ||R_k||^2 partials ----
    double rn = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (hi || t < NT2) {
          double v = Rr[t][r];
          Rg[(size_t)(16 * (my_t0 + t) + mr * r + mg * g) * 16 + col] = v;
          rn += v * v;
        }
      }
    Lrn[mw * 4 + g][col] = rn;
    __syncthreads();
  }
  // ---- per-trait sums ----
  if (tid < 16) {
    int k2 = tile * 16 + tid;
    double r2 = 0.0;
    for (int s = 0; s < NWM * 4; s++) r2 += Lrn[s][tid];
    double *sm = a.sums + (size_t)seg_slot * 5 * a.q_pad;
    sm[(size_t)4 * a.q_pad + k2] = r2;
    for (int v = 0; v < 4; v++) {
      double acc2 = 0.0;
      for (int jj = 0; jj < 16; jj++) acc2 += Lred[v][jj * 16 + tid];
      sm[(size_t)v * a.q_pad + k2] = acc2;
    }
  }
  if (SEG) {
    // publish this tile's residual: every storing wave drains its stores, then one agent-scope release + flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.done[tile], seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
