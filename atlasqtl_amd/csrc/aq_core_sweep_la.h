// aq_core_sweep_la.h -- core sweep kernel, "look-ahead" form (the one bench.py measures).
//
// Blocked Gauss-Seidel in n-space for src/coreLoop.cpp:38-86 (see aq_core_sweep.h), arranged so that the sequential
// 16-SNP pass -- a chain of ~21 dependent fp64 operations per SNP that cannot be shortened -- runs CONCURRENTLY with the
// matrix work:
//
//   workgroup = 8 waves for TT in {1, 2} tiles of 16 traits:
//       6 "matrix" waves (0,1,2 and 4,5,6: two per SIMD on three SIMDs), each owning NT (waves 0-2) or NT2 (waves 4-6)
//         16-sample tiles of the residual R_K of every trait tile in VGPRs,
//       1 "recurrence" wave (wave 3; lane = (row group, trait): 4 / TT groups of 16 TT traits),
//       1 "helper" wave (wave 7, same SIMD as wave 3): stages gam, mu, the Gram blocks and the probit terms of the next
//         SNP block, finalises the previous one (stores, column / row sums).
//     fp64 VALU work and f64 MFMAs share one datapath (a chain that has to wait behind MFMAs is 2.6x slower), so all fp64
//     VALU work lives on SIMD 3.  With two trait tiles per workgroup the recurrence wave also owns NT3 = 3 or 6 residual
//     tiles (aq_la_nt3) whose matrix work it does BETWEEN two chains -- SIMD 3's MFMA slots would otherwise go unused.
//   phase b:
//     recurrence wave : SNP block b.   s_j = S'_b[j] - (X_b'X_{b-1} delta_{b-1})[j]      cross-block Gram, precomputed
//                                            - sum_{i<j} (X_b'X_b)[j,i] delta_i            in-block Gram
//                       then mu, gam, m1, delta_j as src/coreLoop.cpp:69-79
//     matrix waves    : R_K -= X_{b-1} delta_{b-1}   (update of the block finished one phase ago)
//                       S'_{b+1} = X_{b+1}' R_K        (f64 MFMA, k = samples)
//   S'_{b+1} misses only the update of block b, which the recurrence wave adds as the 16x16 cross-Gram correction, so
//   the result is the same Gauss-Seidel sweep.
//
// TT = 2 (two trait tiles per workgroup, used when there are enough tiles to fill the chip): every X operand fetched
// from L2 feeds two MFMAs, the chain is evaluated once for 32 traits (two row groups instead of four: 37 % fewer fp64
// VALU instructions per trait), and a phase carries twice the matrix work, so the fixed per-phase costs (hand-off
// counters, delta read, accumulator drain, S' store) weigh half as much.  The two matrix waves of a SIMD run out of step
// (waves 4-6 start a phase when their partner is a given number of tiles into it: a.stagger); that paid while the matrix
// SIMDs were the bound, with tiles on the recurrence wave SIMD 3 is co-critical and any stagger 0 .. 9 measures the same.
//
// The X operand stream is issued by inline-asm loads with explicit s_waitcnt vmcnt counts: the compiler can neither
// hoist the loads of a fully unrolled tile loop (which cost 16 VGPRs per residual tile and spilled) nor has it to
// guess the prefetch distance.  LDS buffers are double-buffered by block parity; the hand-offs are point-to-point LDS
// progress counters (no workgroup barrier inside the sweep).
#pragma once
#include <hip/hip_runtime.h>
#include "aq_core_sweep.h"
#include <type_traits>

typedef double aq_v2 __attribute__((ext_vector_type(2)));

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane: quad butterflies, then row rotations by
// 4 and 8 (every quad holds its own sum by then).  Eight v_mov_b32 DPP + four v_add_f64 with register latency, where
// __shfl_xor costs eight ds_bpermute_b32 round trips through the LDS pipeline.
__device__ __forceinline__ double aq_row16_sum(double v) {
  auto step = [&](auto ctrl) __attribute__((always_inline)) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xf, 0xf, false);
    v += __hiloint2double(hi, lo);
  };
  step(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  step(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  step(std::integral_constant<int, 0x124>{});   // row_ror:4
  step(std::integral_constant<int, 0x128>{});   // row_ror:8
  return v;
}

// cache policy of the Gram-block stream (49 KB per phase and workgroup, read once per sweep): 2 = nt, non-temporal, so that it
// does not push the X operand panels -- which every workgroup of the XCD re-reads -- out of the L2
#ifndef AQ_GLDS_AUX
#define AQ_GLDS_AUX 2
#endif
// One LDS-DMA transfer (global_load_lds_dwordx4): 64 lanes x 16 B from per-lane global addresses to lds_base + 16 lane, with
// no VGPR destination; completion is counted in vmcnt.  lds_base must be wave-uniform.
__device__ __forceinline__ void aq_glds16(const void *gsrc_lane, void *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, AQ_GLDS_AUX);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The residual tiles live in
// registers only while every index into them is a constant expression.
template <int... Is, class F>
__device__ __forceinline__ void aq_static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void aq_static_for(F &&f) {
  aq_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// two 16-byte loads (one operand tile = 2 x 1 KB per wave): lane address = sbase + voff (+ 1024).  The base is wave-uniform
// but computed inside the role split, where the compiler no longer proves it: v_readfirstlane puts it into the SGPR pair the
// saddr form needs (folded away wherever uniformity is known).
#define AQ_LD2(d0, d1, voff, sbase)                                                                            \
  do {                                                                                                          \
    const unsigned long long b_ = (unsigned long long)(sbase);                                                  \
    const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)b_);                                          \
    const unsigned hi_ = __builtin_amdgcn_readfirstlane((unsigned)(b_ >> 32));                                  \
    const unsigned long long u_ = ((unsigned long long)hi_ << 32) | lo_;                                        \
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"                \
                 : "=&v"(d0), "=&v"(d1)                                                                         \
                 : "v"(voff), "s"(u_));                                                                         \
  } while (0)
// the same with a compile-time byte offset IMM (and IMM + 1024) in the instruction: one SGPR base serves four tiles
// (IMM in {-4096, -2048, 0, 2048}; the 13-bit signed offset field holds -4096 .. 4095)
#define AQ_LD2I(d0, d1, voff, sbase, IMM)                                                                      \
  do {                                                                                                          \
    const unsigned long long b_ = (unsigned long long)(sbase);                                                  \
    const unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)b_);                                          \
    const unsigned hi_ = __builtin_amdgcn_readfirstlane((unsigned)(b_ >> 32));                                  \
    const unsigned long long u_ = ((unsigned long long)hi_ << 32) | lo_;                                        \
    asm volatile("global_load_dwordx4 %0, %2, %3 offset:%4\n\tglobal_load_dwordx4 %1, %2, %3 offset:%5"        \
                 : "=&v"(d0), "=&v"(d1)                                                                         \
                 : "v"(voff), "s"(u_), "i"(IMM), "i"((IMM) + 1024));                                            \
  } while (0)
// wait until at most N of this wave's vector-memory operations are outstanding; the operands tie the wait to the
// registers it covers so that no use can be scheduled above it
#define AQ_WAIT2(N, d0, d1) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(d0), "+v"(d1))
template <int N>
__device__ __forceinline__ void aq_wait2n(aq_v2 &d0, aq_v2 &d1) {   // the same with a count computed at compile time
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(d0), "+v"(d1) : "n"(N));
}

// Operand prefetch with D buffers per stream (tile k in buffer k % D).  Request order of a phase: the tiles left dangling by the
// phase before -- [XU0 XA0 XU1 XA1 ...], XU up to tile D-2, XA up to D-3 -- then per step t: XU(t+D-1), XA(t+D-2) while they
// exist.  aq_req_index = position of a request in that order, aq_req_issued = requests out after the issues of step t; a wait
// for a tile is vmcnt(2 x (requests issued after it)): two loads per request, completed in order.
constexpr int aq_req_index(int D, int last, int stream /* 0 XU, 1 XA */, int k) {
  int n = 0;
  for (int j = 0; j <= D - 2; j++) {
    if (stream == 0 && k == j) return n;
    n++;
    if (j <= D - 3) {
      if (stream == 1 && k == j) return n;
      n++;
    }
  }
  for (int t = 0; t <= last; t++) {
    if (t + D - 1 <= last) {
      if (stream == 0 && k == t + D - 1) return n;
      n++;
    }
    if (t + D - 2 <= last) {
      if (stream == 1 && k == t + D - 2) return n;
      n++;
    }
  }
  return -1;
}
constexpr int aq_req_issued(int D, int last, int t) {
  int n = (D - 1) + (D - 2);
  for (int s = 0; s <= t; s++) n += (s + D - 1 <= last) + (s + D - 2 <= last);
  return n;
}
#ifndef AQ_DEEP_TT1
#define AQ_DEEP_TT1 4   // buffers per operand stream of the one-tile workgroups
#endif

// NT = residual tiles of matrix waves 0,1,2; NT2 (= NT or NT-1) those of waves 4,5,6: each SIMD carries NT + NT2.
// SEG: chained-segment launch (a.nseg * nwg workgroups).  Workgroup s*nwg + k handles SNP segment s of trait-tile group
// k, starting from the residual that segment s-1 of the same group left in global memory.  Blocks are dispatched in
// index order, so that workgroup has normally finished long before; correctness does not depend on it: the hand-off
// is an agent-scope release (producer) / acquire (consumer) around done[k], and the wait is bounded.
// MASK: Y with missing values (reference coreDualMisLoop, src/coreLoop.cpp:91-138): masked residual, per-trait Gram blocks and
// per-entry sig2_beta_vb, the NA forms of the column sums (six rows).  One trait tile per workgroup only.
template <int NT, int NT2, bool SEG, int TT, bool MASK = false, int NT3_ = -1>
__global__ __launch_bounds__(8 * 64, 2) void aq_core_sweep_la_kernel(const AqCoreArgs a) {
  static_assert(!MASK || TT == 1, "the masked form keeps 16 per-trait Gram blocks in LDS: one trait tile per workgroup");
  constexpr int NWM = 6;                        // matrix waves: 0,1,2,4,5,6
  // residual tiles of the recurrence wave (its matrix work follows its chain): aq_la_nt3 unless the instance names its own count
  constexpr int NT3 = NT3_ >= 0 ? NT3_ : aq_la_nt3(NT, NT2, TT);
  constexpr int NPS = NWM + (NT3 > 0 ? 1 : 0);  // partial S' slots
  constexpr int NTR = 16 * TT;                  // traits per workgroup
  constexpr int ENT = 256 * TT;                 // entries of one SNP block: [snp][trait]
  constexpr int NG = 4 / TT;                    // 16*TT-lane groups of the recurrence / helper wave
  constexpr int RPG = 16 / NG;                  // SNP rows per group
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const bool is_rec = (w == 3);
  const int mw = w < 3 ? w : w - 1;             // matrix-wave index 0..5
  const int g = lane >> 4;
  const int col = lane & 15;
  const int nwg = a.ntile / TT;                 // the host pads q so that ntile is a multiple of TT
  int wg_ = blockIdx.x, seg_b0 = 0, seg_b1 = a.nb, seg = 0;
  if (SEG) {
    seg = blockIdx.x / nwg;
    wg_ = blockIdx.x - seg * nwg;
    seg_b0 = (int)((long long)a.nb * seg / a.nseg);
    seg_b1 = (int)((long long)a.nb * (seg + 1) / a.nseg);
    if (seg > 0) {
      if (tid == 0) {
        int tries = 0;
        while (__hip_atomic_load(&a.done[wg_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg) {
          __builtin_amdgcn_s_sleep(32);
          if (++tries > 4000000) { *a.errflag = 1; break; }   // bounded: never hang the GPU
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
  }
  // sample split (C > 1, never together with SEG): workgroup k*C + part = part of trait group k -- partners are adjacent in
  // dispatch order, so a waiting part always has its partners resident or next to be dispatched
  const int C = SEG ? 1 : a.C;
  const int part = C > 1 ? wg_ % C : 0;
  const int wg = C > 1 ? wg_ / C : wg_;
  const bool lead = (part == 0);                // the part that records the group's results
  const int tile0 = wg * TT;                    // first 16-trait tile of this workgroup
  const int nblk = seg_b1 - seg_b0;
  const bool helper = (w == 7);
  // recurrence / helper lane map: trait slot ht = lane % NTR (tile ht >> 4, column ht & 15), row group hg = lane / NTR;
  // entry (snp j, trait ht) of a block lives at j * NTR + ht in the LDS arrays; group hg owns the rows hg + NG r
  const int ht = lane & (NTR - 1), hg = lane / NTR;

  __shared__ double Sp[2][NPS][ENT];   // partial S' of each wave that owns residual tiles [snp][trait]
  __shared__ double LA[2][ENT];        // A = log(1-Phi) - log Phi
  __shared__ double Lm1[2][ENT];       // old m1 = gam*mu
  __shared__ double LB[2][ENT];        // slope b of Z
  __shared__ double Laa[2][ENT];       // intercept a of Z (Z = a + gam b, R/update_vb.R:217-234)
  __shared__ double LG[MASK ? 1 : 2][MASK ? 1 : 512];    // X_b'X_b as [16][32], upper 16 columns zero        (complete Y)
  // Complete Y, where the cross-block correction X_b'X_{b-1} delta_{b-1} comes from: with two trait tiles per workgroup the matrix
  // SIMDs are the bound and the helper wave forms the product after each chain (Lcx); with ONE tile per workgroup (the trait
  // shards of a multi-GPU run, small q: the chain is the critical path) it is accumulated INSIDE the chain of block b-1, one
  // column of X_b'X_{b-1} per step as that step's delta appears -- complete the moment that chain ends (CXC).  The chain of
  // block b-1 then needs block b's cross Gram: three buffers, block m in LGx[m % 3], staged a block earlier.
  constexpr bool CXC = (TT == 1) && !MASK;
  __shared__ double LGx[MASK ? 1 : 3][MASK ? 1 : 256];   // X_b'X_{b-1}  [j][i]                              (complete Y)
  // MASK: the traits' own blocks, written by LDS-DMA.  Diagonal block: lower triangle [i (i + 1) / 2 + j][trait], by block
  // parity; cross block [j][i][trait], ONE buffer (released by the recurrence wave through Fl[13] as soon as the correction
  // at the start of a chain has read it).  Per-entry constants of the chain.
  __shared__ double LGk2[MASK ? 2 : 1][MASK ? AQ_GK_DIAG : 1];
  __shared__ double LGxk[MASK ? 4096 : 1];
  __shared__ double Lcoef[2][MASK ? ENT : 1], LK[2][MASK ? ENT : 1], Ls2[2][MASK ? ENT : 1], Lls2[2][MASK ? ENT : 1], Lxn[2][MASK ? ENT : 1];
  __shared__ double Lgam[2][ENT], Lmu[2][ENT], Ldel[2][ENT];
  __shared__ double Lcx[MASK ? 1 : ENT];   // complete Y: X_b'X_{b-1} delta_{b-1} of the block whose chain comes next, from the helper wave
  __shared__ double Ldum[ENT];         // where the recurrence wave's lanes of row groups > 0 put their (identical) gam, mu, delta
  __shared__ double Stot[(SEG || TT == 2) ? 1 : 2][(SEG || TT == 2) ? 1 : ENT];   // sample split: S' of a block summed over all parts (from the helper wave)
  __shared__ double Lpt[AQ_PT_LEN];    // the probit tables (aq_probit_tab.h), [function][coefficient][interval]: 6.2 KB
  __shared__ double Lred[6][64];       // the helper lanes' column sums [row group][trait], added up per trait at the end
  __shared__ double Lrn[NPS * 4][NTR];
  // Point-to-point progress counters instead of a workgroup barrier per phase: Fl[0..5] = number of SNP blocks whose
  // partial S' matrix wave m has written, Fl[6] = blocks the recurrence wave has finished, Fl[7] = blocks the helper
  // wave has staged, Fl[8..10] = phases that matrix wave 0..2 has carried past its stagger tile (its SIMD partner 4..6
  // starts the phase then), Fl[11] = phases the recurrence wave's own matrix work has finished (init mode: the helper may
  // then reuse a beta buffer), Fl[12] = (complete Y) blocks whose cross-block correction the helper wave has put into Lcx, Fl[13] = (MASK) chains that have read their cross blocks, Fl[14] = (sample split) blocks whose
  // S' the helper wave has exchanged with the other parts.  Each wave waits only for what it really reads, so the matrix waves -- the critical path --
  // never stop at a barrier.  LDS operations of a wave execute in order and the LDS is one pipeline per CU, so
  // "data stores; s_waitcnt lgkmcnt(0); counter store" on one side and "counter load ... ; data loads" on the other are
  // ordered; the asm memory clobbers keep the compiler from moving accesses across them.
  __shared__ __attribute__((aligned(16))) int Fl[16];
  // (explicit LDS address space: through a generic pointer the volatile accesses become flat loads with a vmcnt(0) drain)
  typedef __attribute__((address_space(3))) volatile int aq_lds_vint;
  aq_lds_vint *Flv = (aq_lds_vint *)(__attribute__((address_space(3))) int *)Fl;
  auto signal = [&](int idx, int val) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) Flv[idx] = val;
  };
#ifdef AQ_DIAG_TIME
  long long dg_wait = 0, dg_wait_b = 0;   // waits on the matrix / recurrence counters; on the helper's and the stagger counters
  const long long dg_t0 = __builtin_readcyclecounter();
#endif
#if defined(AQ_DIAG_TIME) || defined(AQ_DIAG_TL)   // (AQ_DIAG_TL: the timeline marks alone -- a lighter build for instances whose
                                                    // registers the wait counters would push over the edge)
  // timeline of workgroup 0, phases 64 .. 95: slot k of (wave, phase) <- cycle counter (behind the per-wave counters in a.dbg)
  auto tl_mark = [&](int i, int k) __attribute__((always_inline)) {
    if (a.dbg && blockIdx.x == 0 && i >= 64 && i < 96 && lane == 0)
      a.dbg[(size_t)gridDim.x * 24 + ((size_t)w * 32 + (i - 64)) * 8 + k] = __builtin_readcyclecounter();
  };
#else
  auto tl_mark = [&](int, int) __attribute__((always_inline)) {};
#endif
  auto wait_ge = [&](int idx, int val) __attribute__((always_inline)) {
#ifdef AQ_DIAG_TIME
    const long long t_in = __builtin_readcyclecounter();
#endif
    while (Flv[idx] < val) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
#ifdef AQ_DIAG_TIME
    if (idx >= 7) dg_wait_b += __builtin_readcyclecounter() - t_in;
    else dg_wait += __builtin_readcyclecounter() - t_in;
#endif
  };
  // the recurrence wave's wait at the top of a chain -- all six matrix waves' partial S' and the helper's staging -- as ONE poll of
  // two 16-byte LDS reads instead of seven round trips in a row (0.3 us per SNP block where the chain is the critical path)
  typedef int aq_i4 __attribute__((ext_vector_type(4)));
  auto wait_s_and_staged = [&](int val) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) volatile aq_i4 aq_lds_vi4;
    aq_lds_vi4 *fp = (aq_lds_vi4 *)(__attribute__((address_space(3))) int *)Fl;
#ifdef AQ_DIAG_TIME
    const long long t_in = __builtin_readcyclecounter();
#endif
    for (;;) {
      const aq_i4 f0 = fp[0], f1 = fp[1];   // Fl[0..3], Fl[4..7]; Fl[6] is this wave's own counter
      const int lo = min(min(min(f0.x, f0.y), min(f0.z, f0.w)), min(min(f1.x, f1.y), f1.w));
      if (lo >= val) break;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
#ifdef AQ_DIAG_TIME
    dg_wait += __builtin_readcyclecounter() - t_in;
#endif
  };
  if (tid < 16) Fl[tid] = 0;

  const int ktrait = tile0 * 16 + ht;           // helper / recurrence: this lane's trait
  const bool kvalid = ktrait < a.q;
  double sig2b_k = 1.0;
  if (helper) {
    sig2b_k = a.sig2b[ktrait];
    if (a.mode != 1)
      for (int e = lane; e < AQ_PT_LEN; e += 64) Lpt[e] = aq_pt_dev[e];
#pragma unroll
    for (int r = 0; r < RPG; r++) {
      const int e = lane + 64 * r;
      Ldel[0][e] = Ldel[1][e] = 0.0;   // read (times zero) by the matrix waves' first two phases
    }
    if constexpr (!MASK) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int e = lane + 64 * r;
        LG[0][(e >> 4) * 32 + 16 + (e & 15)] = 0.0;
        LG[1][(e >> 4) * 32 + 16 + (e & 15)] = 0.0;
      }
    }
  }

  // One code path for both kinds of matrix wave and both modes.  Phase i = 0 .. nblk + 1 of a segment starting at block b0:
  //     update with block b0 + i - 2 (i >= 2; its delta must be out: recurrence counter >= i - 1)
  //     S' of block b0 + i          (i < nblk, sweep mode only)           then announce i + 1 on this wave's counter.
  // Phases without an update (or without S') run the same instruction stream with a zero delta (or drop the accumulator):
  // 4 of the nblk + 2 phases of a segment do some idle MFMAs, and in exchange the tile loop exists once, without a branch.
  // ROLE 0: matrix waves 0-2 (NT tiles each), 1: matrix waves 4-6 (NT2 each), 2: the recurrence wave (NT3 tiles; `hook(i)` runs
  // its chain for block b0 + i - 1 at the top of phase i -- that block's S' is complete when every matrix wave has announced
  // phase i - 1, and its own share of it was stored by this very wave at the end of its previous phase).
  constexpr int NTT = 3 * (NT + NT2) + NT3;
  const int nblk_s = a.mode == 1 ? 0 : nblk;       // init mode: no S'
  auto run = [&](auto ntc, auto rolec, auto hook) __attribute__((always_inline)) {
    constexpr int NTC = decltype(ntc)::value;
    constexpr int ROLE = decltype(rolec)::value;
    constexpr bool HI = ROLE == 0;
    constexpr int ST = (NTC + 2) / 3;              // stagger: the SIMD partner enters the phase after this many tiles
    const int my_t0 = part * NTT + (ROLE == 0 ? mw * NT : ROLE == 1 ? 3 * NT + (mw - 3) * NT2 : 3 * (NT + NT2));
    const int slot = ROLE == 2 ? NWM : mw;         // partial S' slot; progress counter: Fl[mw], the recurrence wave's is Fl[11]
    // residual tiles: Rr[tt][t][r] <-> sample 16*(my_t0+t) + 4 r + g, trait col of tile tile0 + tt   (f64 MFMA D layout
    // row = 4 reg + (lane >> 4): the host refuses to run this kernel on a device that reports the other map)
    aq_d4 Rr[TT][NTC];
    unsigned long long mb0 = 0, mb1 = 0;   // MASK: bit 4 t + r = entry (tile t, register r) observed; tiles 16, 17 in mb1
    aq_static_for<TT>([&](auto ttc) __attribute__((always_inline)) {
      constexpr int tt = decltype(ttc)::value;
      const double *Rg = a.R + (size_t)(tile0 + tt) * a.n_pad * 16 + (size_t)(16 * my_t0 + g) * 16 + col;
      aq_static_for<NTC>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int r = 0; r < 4; r++) Rr[tt][t][r] = Rg[(16 * t + 4 * r) * 16];
        if constexpr (MASK) {
          const double *Mg = a.mis + (size_t)tile0 * a.n_pad * 16 + (size_t)(16 * my_t0 + g) * 16 + col;
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (Mg[(16 * t + 4 * r) * 16] != 0.0) { if (t < 16) mb0 |= 1ull << (4 * (t & 15) + r); else mb1 |= 1ull << (4 * (t & 15) + r); }
        }
      });
    });
    auto remask = [&](auto tc) __attribute__((always_inline)) {   // R_K = mis .* R_K for one residual tile (R/...core.R:19-22 in n-space)
      constexpr int t = decltype(tc)::value;
      const unsigned long long mb = t < 16 ? mb0 : mb1;
#pragma unroll
      for (int r = 0; r < 4; r++) Rr[0][t][r] = ((mb >> (4 * (t & 15) + r)) & 1ull) ? Rr[0][t][r] : 0.0;
    };
    // X operand streams: [nb][NTT][2][64] x 16 B; this wave's tiles start at my_t0, lane address = block base + voff
    const unsigned voff = (unsigned)((my_t0 * 128 + lane) * 16);
    const char *XUb = (const char *)a.XU, *XAb = (const char *)a.XA;
    const long long BLK = (long long)NTT * C * 128 * 16;   // bytes per SNP block (all parts)
    aq_v2 p0, p1, q0, q1, c0, c1, d0, d1;   // XU tiles alternate between (p0,p1) and (q0,q1), XA tiles between (c0,c1) and (d0,d1)
    // DEEP (above all one trait tile per workgroup: a tile step is only 8 MFMAs = 0.21 us, less than an L2 hit, and there are
    // registers to spare): DB = 3 or 4 buffers per stream instead of two, every request one or two steps earlier -- XU(t+DB-1) and
    // XA(t+DB-2) at the start of step t, the first tiles of the NEXT phase at the end of the last step (aq_req_index above).
    // The destination registers of these requests are written long after the asm statement that names them, which the compiler
    // cannot know.  Under register pressure it splits their live ranges between request and wait: it copies the (stale) bits
    // elsewhere, spills them, and hands the physical registers to other values -- which the returning load then overwrites
    // (an LDS read of delta, an address: the GPU fault seen in round 2 with four buffers in a 256-register MASK instance).
    // profiles/r03_isa_forced_deep.txt shows exactly that in <11, 11, SEG, 2> with four buffers forced (v_mov_b64 copies,
    // a scratch_store and ds_read2_b64 into destinations in flight).  Nothing in the source can forbid it, so it is PROVEN
    // ABSENT per build instead: `make` links the library only after tools/check_isa_operands.py has walked the control flow of
    // every instance's shipped ISA, replayed vmcnt over the real instruction stream and found no instruction touching a
    // destination while its request is in flight (profiles/r03_isa_proof.txt).  The depth per instance is then a matter of
    // measurement only: four buffers for complete Y up to 11 tiles per wave (207 - 219 VGPRs), three for complete Y beyond
    // (<= 229) and for the chained MASK instances (232: C3 with NA); the other MASK instances and the two-tile instances keep
    // the two-buffer scheme below (deeper, their register allocation does what is described above and the build stops).
#ifdef AQ_FORCE_DEEP_DB   // tools/isa_probe only: the deep scheme in instances the shipped build denies it (ISA study, never launched)
#ifdef AQ_FORCE_DEEP_TT2
    constexpr bool DEEP = (NTC >= 4);
#else
    constexpr bool DEEP = (TT == 1) && (NTC >= 4);
#endif
    constexpr int DB = !DEEP ? 1 : AQ_FORCE_DEEP_DB;
#elif defined(AQ_FORCE_DEEP_MASK_DB)   // libatlasqtl_hip_deepmask.so (make deepmask): round 2's faulting configuration -- the deep scheme in
    // EVERY MASK instance, AQ_FORCE_DEEP_MASK_DB buffers -- as a library of its own, for the one GPU run of tests/test_gpu_sharded.py under it
    constexpr bool DEEP = (TT == 1) && (NTC >= 4) && (MASK || true);
    constexpr int DB = !DEEP ? 1 : MASK ? AQ_FORCE_DEEP_MASK_DB : (NT <= 11) ? AQ_DEEP_TT1 : 3;
#else
    constexpr bool DEEP = (TT == 1) && (NTC >= 4) && (!MASK || (SEG && NT <= 11));
    constexpr int DB = !DEEP ? 1 : (!MASK && NT <= 11) ? AQ_DEEP_TT1 : 3;
#endif
    aq_v2 xb[DB][2], ab[DB][2];
    // Tile step t: U(t) = 4 TT MFMAs chained on Rr[.][t]; S(t-1) = 4 TT chained on acc, whose B operand Rr[.][t-1] was
    // finished a whole step earlier.  Loads, all issued at the START of a step: XU(t+1) (used one step later) and XA(t)
    // (used by S(t), one and a half steps later); tile 0 of the NEXT phase at the end of the last step.  Queue of
    // outstanding loads at the waits, oldest first, two loads each:
    //   t=0: [XU0 XA0 | XU1]           wait XU0 = vmcnt(4)
    //   t=1: [XA0 XU1 | XU2 XA1]       wait XU1 = vmcnt(4) (XA0 is older: done as well)
    //   t>=2: [XU(t) XA(t-1) | XU(t+1) XA(t)]   wait XU(t) = vmcnt(6), then XA(t-1) = vmcnt(4)
    //   last: [XU(t) XA(t-1) | XA(t)]  vmcnt(4), vmcnt(2) (two tiles only: [XA0 XU1 | XA1] vmcnt(2)); then +XU'0: [XA(t) XU'0]
    //         vmcnt(2); then +XA'0.
    // the dangling requests [XU0 XA0 XU1 XA1 ...] of a phase whose operands start at xu0 / xa0 (tiles 0 .. 3 share one base)
    auto dangling = [&](const char *xu0, const char *xa0, auto skip_first) __attribute__((always_inline)) {
      aq_static_for<DB - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (!(decltype(skip_first)::value && j == 0)) {
          constexpr int kb = (j + 2) / 4, imm = (j - 4 * kb) * 2048;
          AQ_LD2I(xb[j][0], xb[j][1], voff, xu0 + kb * 8192, imm);
        }
        if constexpr (j <= DB - 3) {
          constexpr int kb = (j + 2) / 4, imm = (j - 4 * kb) * 2048;
          AQ_LD2I(ab[j][0], ab[j][1], voff, xa0 + kb * 8192, imm);
        }
      });
    };
    if constexpr (DEEP) {
      dangling(XUb + seg_b0 * BLK, XAb + seg_b0 * BLK, std::false_type{});
    } else {
      AQ_LD2(p0, p1, voff, XUb + seg_b0 * BLK);
      AQ_LD2(c0, c1, voff, XAb + seg_b0 * BLK);
    }
    for (int i = 0; i <= nblk + 1; i++) {
      const bool do_u = i >= 2, do_s = i < nblk_s;
      const double mflag = do_u ? -1.0 : 0.0;
      const int bu = seg_b0 + (i >= 2 ? i - 2 : 0), bs = seg_b0 + (i < nblk ? i : nblk - 1);
      const int nbu = seg_b0 + (i >= 1 ? (i - 1 < nblk ? i - 1 : nblk - 1) : 0), nbs = seg_b0 + (i + 1 < nblk ? i + 1 : nblk - 1);
      tl_mark(i, 0);                     // phase entered (recurrence wave: before its chain)
      hook(i);
      tl_mark(i, 1);                     // (recurrence wave: chain done)
      if (do_u && (ROLE != 2 || a.mode == 1)) wait_ge(6, i - 1);   // (sweep mode: the recurrence wave wrote that delta itself)
      if (ROLE == 1 && a.stagger) wait_ge(8 + (mw - 3), i + 1);
      tl_mark(i, 2);                     // waits passed: tile loop starts
      double nd[TT][4];
      {
        // (4 s + g) NTR + 16 tt + col.  The lane id is recomputed here and hidden from the optimiser: kept in a register
        // across the phase it was spilled in earlier versions, and the reload's s_waitcnt vmcnt(0) drained the operand prefetch
        unsigned zero = 0;
        asm volatile("" : "+v"(zero));
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
        const double *dl = &Ldel[bu & 1][(ln >> 4) * NTR + (ln & 15)];
#pragma unroll
        for (int tt = 0; tt < TT; tt++)
#pragma unroll
          for (int s = 0; s < 4; s++) nd[tt][s] = dl[4 * NTR * s + 16 * tt] * mflag;   // (Ldel starts zeroed: never NaN)
      }
      aq_d4 acc[TT];
#pragma unroll
      for (int tt = 0; tt < TT; tt++) acc[tt] = (aq_d4){0, 0, 0, 0};
      const char *xu = XUb + __builtin_amdgcn_readfirstlane(bu) * BLK, *xa = XAb + __builtin_amdgcn_readfirstlane(bs) * BLK;
      const char *nxu = XUb + __builtin_amdgcn_readfirstlane(nbu) * BLK, *nxa = XAb + __builtin_amdgcn_readfirstlane(nbs) * BLK;
      // Two matrix waves share a SIMD.  At equal priority the one that streams MFMAs wins every issue slot and its partner's
      // hand-off work (counter polls, the delta read, address arithmetic: a few dozen VALU operations) only advances in the
      // stream's gaps -- the two loops ran strictly one after the other, hand-offs included.  So a matrix wave runs its hand-offs
      // at raised priority (they slip between the partner's MFMAs) and drops back for its own MFMA stream.
      if (ROLE != 2 && a.mprio) __builtin_amdgcn_s_setprio(0);
      auto U = [&](auto tc, aq_v2 u0, aq_v2 u1) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        aq_static_for<TT>([&](auto ttc) __attribute__((always_inline)) {
          constexpr int tt = decltype(ttc)::value;
          aq_d4 Rt = Rr[tt][t];
          Rt = aq_mfma(u0.x, nd[tt][0], Rt);
          Rt = aq_mfma(u0.y, nd[tt][1], Rt);
          Rt = aq_mfma(u1.x, nd[tt][2], Rt);
          Rt = aq_mfma(u1.y, nd[tt][3], Rt);
          Rr[tt][t] = Rt;
        });
      };
      auto S = [&](auto tc, aq_v2 x0, aq_v2 x1) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        aq_static_for<TT>([&](auto ttc) __attribute__((always_inline)) {
          constexpr int tt = decltype(ttc)::value;
          const aq_d4 Rp = Rr[tt][t];
          acc[tt] = aq_mfma(x0.x, Rp[0], acc[tt]);
          acc[tt] = aq_mfma(x0.y, Rp[1], acc[tt]);
          acc[tt] = aq_mfma(x1.x, Rp[2], acc[tt]);
          acc[tt] = aq_mfma(x1.y, Rp[3], acc[tt]);
        });
      };
      if constexpr (DEEP) {
        aq_static_for<NTC>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value, last = NTC - 1;
          using TP = std::integral_constant<int, (t > 0 ? t - 1 : 0)>;
          if constexpr (t + DB - 1 <= last) {
            constexpr int k = t + DB - 1, kb = (k + 2) / 4, imm = (k - 4 * kb) * 2048;
            AQ_LD2I(xb[k % DB][0], xb[k % DB][1], voff, xu + kb * 8192, imm);
          }
          if constexpr (t + DB - 2 <= last) {
            constexpr int k = t + DB - 2, kb = (k + 2) / 4, imm = (k - 4 * kb) * 2048;
            AQ_LD2I(ab[k % DB][0], ab[k % DB][1], voff, xa + kb * 8192, imm);
          }
          constexpr int issued = aq_req_issued(DB, last, t);
          aq_wait2n<2 * (issued - 1 - aq_req_index(DB, last, 0, t))>(xb[t % DB][0], xb[t % DB][1]);
          U(tc, xb[t % DB][0], xb[t % DB][1]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (t >= 1) {
            constexpr int k = t - 1;
            aq_wait2n<2 * (issued - 1 - aq_req_index(DB, last, 1, k))>(ab[k % DB][0], ab[k % DB][1]);
            S(TP{}, ab[k % DB][0], ab[k % DB][1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (MASK) {
            remask(tc);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (t == last) {
            AQ_LD2I(xb[0][0], xb[0][1], voff, nxu, 0);          // XU'0: every U of this phase is issued
            aq_wait2n<2 * (issued - 1 - aq_req_index(DB, last, 1, last) + 1)>(ab[last % DB][0], ab[last % DB][1]);   // XA(last)
            S(tc, ab[last % DB][0], ab[last % DB][1]);
            __builtin_amdgcn_sched_barrier(0);
            dangling(nxu, nxa, std::true_type{});               // XA'0, XU'1, ... (XU'0 is out already)
          }
          if constexpr (HI && t + 1 == ST) {
            if (a.stagger) signal(8 + mw, i + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      } else if constexpr (NTC == 1) {
        using C0 = std::integral_constant<int, 0>;
        AQ_WAIT2(0, p0, p1);
        AQ_WAIT2(0, c0, c1);
        U(C0{}, p0, p1);
        if (HI && a.stagger) signal(8 + mw, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        AQ_LD2(p0, p1, voff, nxu);
        if constexpr (MASK) remask(C0{});
        S(C0{}, c0, c1);
        __builtin_amdgcn_sched_barrier(0);
        AQ_LD2(c0, c1, voff, nxa);
      } else {
        aq_static_for<NTC>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value;
          constexpr bool even = (t & 1) == 0, last = (t == NTC - 1);
          using TP = std::integral_constant<int, (t > 0 ? t - 1 : 0)>;   // the tile S' runs behind on
          if constexpr (!last) {
            constexpr int kb = (t + 1 + 2) / 4, imm = (t + 1 - 4 * kb) * 2048;   // tile t+1 relative to base group kb
            if constexpr (even) { AQ_LD2I(q0, q1, voff, xu + kb * 8192, imm); }
            else { AQ_LD2I(p0, p1, voff, xu + kb * 8192, imm); }
          }
          if constexpr (t >= 1) {
            constexpr int kc = (t + 2) / 4, immc = (t - 4 * kc) * 2048;          // tile t of the XA stream
            if constexpr (even) { AQ_LD2I(c0, c1, voff, xa + kc * 8192, immc); }
            else { AQ_LD2I(d0, d1, voff, xa + kc * 8192, immc); }
          }
          // U(t)
          if constexpr (t == 0) AQ_WAIT2(4, p0, p1);
          else if constexpr (last && t == 1) AQ_WAIT2(2, q0, q1);   // two tiles only: [XA0 XU1 | XA1]
          else if constexpr (last) { if constexpr (even) AQ_WAIT2(4, p0, p1); else AQ_WAIT2(4, q0, q1); }
          else if constexpr (t == 1) { AQ_WAIT2(4, q0, q1); AQ_WAIT2(4, c0, c1); }
          else if constexpr (even) AQ_WAIT2(6, p0, p1);
          else AQ_WAIT2(6, q0, q1);
          if constexpr (even) U(tc, p0, p1); else U(tc, q0, q1);
          __builtin_amdgcn_sched_barrier(0);
          // S(t-1): XA(t-1) sits in (d0,d1) for even t, in (c0,c1) for odd t
          if constexpr (t >= 1) {
            if constexpr (last) { if constexpr (even) AQ_WAIT2(2, d0, d1); else AQ_WAIT2(2, c0, c1); }
            else if constexpr (t >= 2) { if constexpr (even) AQ_WAIT2(4, d0, d1); else AQ_WAIT2(4, c0, c1); }
            if constexpr (even) S(TP{}, d0, d1); else S(TP{}, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (MASK) {   // behind S(t-1): the MFMA pipe works on that while U(t) completes
            remask(tc);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (last) {
            AQ_LD2(p0, p1, voff, nxu);       // tile 0 of the next phase; (p0,p1) is free: XU(t) sat there only for even t, and U(t) is issued
            if constexpr (even) { AQ_WAIT2(2, c0, c1); S(tc, c0, c1); }
            else { AQ_WAIT2(2, d0, d1); S(tc, d0, d1); }
            __builtin_amdgcn_sched_barrier(0);
            AQ_LD2(c0, c1, voff, nxa);       // (c0,c1) held XA(t) for even t: S(t) is issued
          }
          if constexpr (HI && t + 1 == ST) {
            if (a.stagger) signal(8 + mw, i + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if (ROLE != 2 && a.mprio) __builtin_amdgcn_s_setprio(2);
      if (do_s) {
        const int ps = bs & 1;
#pragma unroll
        for (int tt = 0; tt < TT; tt++)
#pragma unroll
          for (int r = 0; r < 4; r++) Sp[ps][slot][(4 * r + g) * NTR + 16 * tt + col] = acc[tt][r];
      }
      signal(ROLE == 2 ? 11 : mw, i + 1);
      tl_mark(i, 3);                     // S' stored and announced
    }
    if constexpr (DEEP) {   // the dangling prefetch (the empty statements tie the wait to every buffer)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(ab[0][0]), "+v"(ab[0][1]));
      if constexpr (DB > 1) asm volatile("" : "+v"(xb[DB > 1 ? 1 : 0][0]), "+v"(xb[DB > 1 ? 1 : 0][1]), "+v"(ab[DB > 1 ? 1 : 0][0]), "+v"(ab[DB > 1 ? 1 : 0][1]));
      if constexpr (DB > 2) asm volatile("" : "+v"(xb[DB > 2 ? 2 : 0][0]), "+v"(xb[DB > 2 ? 2 : 0][1]), "+v"(ab[DB > 2 ? 2 : 0][0]), "+v"(ab[DB > 2 ? 2 : 0][1]));
      if constexpr (DB > 3) asm volatile("" : "+v"(xb[DB > 3 ? 3 : 0][0]), "+v"(xb[DB > 3 ? 3 : 0][1]), "+v"(ab[DB > 3 ? 3 : 0][0]), "+v"(ab[DB > 3 ? 3 : 0][1]));
    }
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(c0), "+v"(c1), "+v"(q0), "+v"(q1), "+v"(d0), "+v"(d1));   // the dangling prefetch
    // ---- write the residual back and ||R_k||^2 partials ----
    // (lane id and base pointer are re-derived behind an optimisation barrier: otherwise the addresses computed for
    // the loads at the top are kept alive -- i.e. spilled -- across the whole sweep)
    unsigned zero2 = 0;
    asm volatile("" : "+v"(zero2));
    const int ln2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero2));
    double *Rout = a.R;
    asm volatile("" : "+s"(Rout));
    aq_static_for<TT>([&](auto ttc) __attribute__((always_inline)) {
      constexpr int tt = decltype(ttc)::value;
      double *Rg = Rout + (size_t)(tile0 + tt) * a.n_pad * 16 + (size_t)(16 * my_t0 + (ln2 >> 4)) * 16 + (ln2 & 15);
      double rn = 0.0;
      aq_static_for<NTC>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double v = Rr[tt][t][r];
          Rg[(16 * t + 4 * r) * 16] = v;
          rn += v * v;
        }
      });
      Lrn[slot * 4 + (ln2 >> 4)][16 * tt + (ln2 & 15)] = rn;
    });
  };
  const auto no_hook = [](int) {};

  // ---- sample split (C > 1): publish this part's partial S' of block b (lane = (row group, trait), own[r] = row hg + NG r),
  // collect the others', add in fixed order; on return own[] holds the sum over all parts, the same bits in every part.
  // Run by the recurrence wave at the start of its chain, or -- a.xhelper, chosen by the host when the matrix waves' phase is
  // longer than chain + exchange -- by the helper wave a block ahead.
  // Every exchanged 64-bit word validates itself: its two lowest mantissa bits carry a tag 1..3 that changes with every use of
  // the slot (slots alternate by block parity and start zeroed: tag 0), so there is no flag, no store acknowledgement to wait for
  // and no ordering between words to rely on -- one trip through the level the XCDs share (sc1 atomics) instead of three.  A
  // reader polls the words until they carry the expected tag.  The sum uses the tagged words themselves (this part's included),
  // i.e. partial sums rounded to 2^-50 relative: identical in every part.  A part overwrites the slot of block b with block b+2
  // only after its own exchange of block b+1, i.e. after every other part has published b+1 -- which each does after having read
  // block b -- so a reader can only ever meet the previous or the expected tag.  No release / acquire fence: at agent scope it
  // would write back and invalidate this XCD's whole L2, where the X operand panels live.
  bool split_dead = false;   // a bounded wait on a partner expired (reported through errflag)
  auto split_exchange = [&](int b, double (&own)[RPG], auto meanwhile) __attribute__((always_inline)) {
    const int par = b & 1;
    const unsigned long long tag = 1ull + (unsigned long long)(((b - seg_b0) >> 1) % 3);
    unsigned long long *slot = (unsigned long long *)a.Pbuf + ((size_t)(wg * 2 + par) * C) * ENT + lane;
    unsigned long long mine[RPG];
#pragma unroll
    for (int r = 0; r < RPG; r++) {
      mine[r] = ((unsigned long long)__double_as_longlong(own[r]) & ~3ull) | tag;
      __hip_atomic_store(&slot[(size_t)part * ENT + 64 * r], mine[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    meanwhile();   // work of the caller that does not depend on the sum, while the words travel
    double tot[RPG];
#pragma unroll
    for (int r = 0; r < RPG; r++) tot[r] = 0.0;
    // partners in ascending order (the fixed order of the sum), the next partner's words requested before the current one's
    // are checked: with everything already published the whole collection costs one trip, not one per partner
    unsigned long long cur[RPG], nxt[RPG];
    auto request = [&](int c2, unsigned long long (&dst)[RPG]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RPG; r++)
        dst[r] = __hip_atomic_load(&slot[(size_t)c2 * ENT + 64 * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    request(part == 0 ? 1 : 0, cur);
    for (int c2 = 0; c2 < C; c2++) {
      if (c2 == part) {
#pragma unroll
        for (int r = 0; r < RPG; r++) tot[r] += __longlong_as_double((long long)mine[r]);
        continue;
      }
      int cn = c2 + 1;
      if (cn == part) cn++;
      if (cn < C) request(cn, nxt);
      int spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < RPG; r++) ok = ok && ((cur[r] & 3ull) == tag);
        if (__all(ok) || split_dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *a.errflag = 1; split_dead = true; }   // give up for good: results are invalid
        request(c2, cur);
      }
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        tot[r] += __longlong_as_double((long long)cur[r]);
        cur[r] = nxt[r];
      }
    }
#pragma unroll
    for (int r = 0; r < RPG; r++) own[r] = tot[r];
  };

  __syncthreads();   // counters and the helper's LDS initialisation are visible to every role
  if (is_rec) {
    // =========================== recurrence wave ===========================================
    if (a.mode != 1) {   // (init mode: the helper wave hands beta to the matrix waves, nothing to do here)
      const double rc_coef = MASK ? 0.0 : a.coef[ktrait];
      const double rc_cinv2s = MASK ? 0.0 : a.c * a.inv2s[ktrait];
      const double rc_K = rc_coef * rc_coef * rc_cinv2s;   // keeps mu off the dependency chain of the recursion
      double cxn[CXC ? RPG : 1];   // (CXC) the next block's cross-block correction, built up by the chain that is running
      auto chain_block = [&](int b) __attribute__((always_inline)) {
        const int par = b & 1;
        // ---- SNP block b.  lane = (hg, ht): ht = trait, hg one of NG row groups -------------------------------------
        // This wave and the helper wave share one SIMD and both are bound by its fp64 VALU issue rate (a dependent
        // v_fma_f64 has a latency of only 6 cycles, tools/microbench/f64_valu.hip), so everything that does not belong
        // to the chain itself is spread over the row groups instead of being repeated in each of them:
        // group hg owns the rows hg, hg + NG, ... of S.
        double Sown[RPG];
        const int need = b - seg_b0 + 1;
        if (C > 1 && a.xhelper) {
          // sample split: the helper wave has added this part's partial S' to the other parts' (a block ahead, off this
          // wave's critical path) -- every part reads the same bits and runs the same chain
          wait_ge(14, need);
          wait_ge(7, need);
#pragma unroll
          for (int r = 0; r < RPG; r++) Sown[r] = Stot[par][(hg + NG * r) * NTR + ht];
        } else {
          // block b needs its six partial S' and its staged scalars
          wait_s_and_staged(need);
#pragma unroll
          for (int r = 0; r < RPG; r++) {
            const int j = hg + NG * r;
            double sv = Sp[par][0][j * NTR + ht];
#pragma unroll
            for (int ww = 1; ww < NPS; ww++) sv += Sp[par][ww][j * NTR + ht];
            Sown[r] = sv;
          }
        }
        tl_mark(need, 4);                  // (diag) partial S' complete and summed
        // cross-block correction X_b'X_{b-1} delta_{b-1} of this group's rows (a segment starts from a complete residual);
        // MASK: with the trait's own cross block X_b' diag(mis_k) X_{b-1}.  It does not depend on S': with the sample split it
        // is computed while the partial sums travel.
        // `correction(sub)`: sub = true subtracts straight from Sown (the form the two-tile instances need: no extra registers),
        // sub = false leaves the terms in cxs[] for later
        double cxs[(!SEG && TT == 1) ? RPG : 1];
        auto correction = [&](auto subc) __attribute__((always_inline)) {
          constexpr bool sub = decltype(subc)::value;
          if constexpr (CXC) {
            // one tile per workgroup: accumulated by the previous chain, in registers
#pragma unroll
            for (int r = 0; r < RPG; r++) {
              const double cx = b > seg_b0 ? cxn[r] : 0.0;
              if constexpr (sub) Sown[r] -= cx;
              else cxs[r] = cx;
            }
          } else if constexpr (!MASK) {
            // complete Y: the helper wave formed the 16 x 16 product right after the previous chain (Fl[12]); here it is 8 reads
            if (b > seg_b0) {
              wait_ge(12, need);
#pragma unroll
              for (int r = 0; r < RPG; r++) {
                const double cx = Lcx[lane + 64 * r];
                if constexpr (sub) Sown[r] -= cx;
                else cxs[r] = cx;
              }
            } else if constexpr (!sub) {
#pragma unroll
              for (int r = 0; r < RPG; r++) cxs[r] = 0.0;
            }
          } else if (b > seg_b0) {
            double dlp[16];
#pragma unroll
            for (int i = 0; i < 16; i++) dlp[i] = Ldel[par ^ 1][i * NTR + ht];
#pragma unroll
            for (int r = 0; r < RPG; r++) {
              double cx = 0.0;
              const double *gx = &LGxk[((hg + NG * r) * 16) * 16 + ht];
#pragma unroll
              for (int i = 0; i < 16; i++) cx += gx[i * 16] * dlp[i];
              if constexpr (sub) Sown[r] -= cx;
              else cxs[r] = cx;
            }
          } else if constexpr (!sub) {
#pragma unroll
            for (int r = 0; r < RPG; r++) cxs[r] = 0.0;
          }
        };
        bool corrected = false;
        if constexpr (!SEG && TT == 1) {
          if (C > 1 && !a.xhelper) {
            split_exchange(b, Sown, [&]() __attribute__((always_inline)) { correction(std::false_type{}); });
#pragma unroll
            for (int r = 0; r < RPG; r++) Sown[r] -= cxs[r];
            corrected = true;
          }
        }
        if (!corrected) correction(std::true_type{});
        tl_mark(need, 5);                  // (diag) correction applied: the 16 steps start
        double sb = __shfl(Sown[0], ht, 64);     // S of SNP 0 (group 0) to every group
        // The 16 steps.  What one step costs is its dependent chain s -> x -> sigmoid -> delta (about 26 fp64 operations) --
        // provided nothing else sits on it.  Two things did (ISA of round 2: the ds_bpermute that fetches the next SNP's S from
        // its owner group was issued at the END of a step, so every step also paid an LDS round trip; and the three stores of
        // gam, mu, delta sat behind an exec-mask branch):
        //   * the S of SNP j+2 is requested at the tail of step j, right after delta_j has been applied to the one register that
        //     holds it, and pinned there (sched_barrier): its round trip runs under the whole of step j+1;
        //   * every lane stores -- the lanes of the other row groups into a scratch area of the same shape (no branch).
        double s_nx = __shfl(Sown[1 / NG], (1 % NG) * NTR + ht, 64);   // S of SNP 1, no in-block update yet
        typedef __attribute__((address_space(3))) double aq_lds_double;
        const bool wsel = lane < NTR;
        aq_lds_double *wdum = (aq_lds_double *)Ldum + ht;
        aq_lds_double *wgam = wsel ? (aq_lds_double *)Lgam[par] + ht : wdum;
        aq_lds_double *wmu = wsel ? (aq_lds_double *)Lmu[par] + ht : wdum;
        aq_lds_double *wdel = wsel ? (aq_lds_double *)Ldel[par] + ht : wdum;
        if constexpr (MASK) {
          signal(13, b - seg_b0 + 1);              // the single cross-block buffer is free for block b+1
          // ---- the same chain with the trait's own diagonal block (lower triangle in LGk) and per-entry sig2_beta_vb
          // (src/coreLoop.cpp:115-133): coef, K and cA come per entry from the helper
          const double *LGk = &LGk2[par][0];
          int rtri[RPG], rrow[RPG];
#pragma unroll
          for (int r = 0; r < RPG; r++) { rrow[r] = hg + NG * r; rtri[r] = rrow[r] * (rrow[r] + 1) / 2; }
          double m1o = Lm1[par][ht], cA = LA[par][ht], cf = Lcoef[par][ht], ci = LK[par][ht], dj = LGk[ht];
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int jn = (j + 1) & 15, j2 = j + 2;
            const int en = jn * NTR + ht;
            double m1o_n = Lm1[par][en], cA_n = LA[par][en], cf_n = Lcoef[par][en], ci_n = LK[par][en];
            double d_n = LGk[(jn * (jn + 1) / 2 + jn) * 16 + ht];
            const double g_next = LGk[(j < 15 ? (jn * (jn + 1) / 2 + j) : 0) * 16 + ht];   // G^(k)[j+1][j]
            const int r2 = (j2 & 15) / NG;
            double gl[RPG];                                   // this group's column j of G^(k), for the in-block update below
#pragma unroll
            for (int r = (j / NG); r < RPG; r++) gl[r] = LGk[(rtri[r] + (rrow[r] < j ? rrow[r] : j)) * 16 + ht];
            __builtin_amdgcn_sched_barrier(0);                // every LDS read that does not depend on this step's delta is on its way
            double s = sb + m1o * dj;                         // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1 (cp_X - cp_X_rm_k)(j,j))   :121
            double mu = cf * s;                               // :125
            double x = fma(-(s * s), ci, cA);                 // :127-129 with mu^2 = coef^2 s^2 (keeps mu off the chain)
            double gm = aq_sigmoid_neg_fast(x);
            double dl = gm * mu - m1o;                        // m1 - m1_old   :130
            sb = s_nx - g_next * dl;                          // S of SNP j+1, complete
            // in-block part of :132, this group's rows (rows <= j are consumed: any finite factor will do for them): first
            // the register that holds SNP j+2, whose S goes on its way to every group at once
            auto upd = [&](int r) __attribute__((always_inline)) { Sown[r] -= gl[r] * dl; };
            if (j2 < 16) {
              Sown[r2] -= gl[r2] * dl;
              s_nx = __shfl(Sown[r2], (j2 % NG) * NTR + ht, 64);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = (j / NG); r < RPG; r++)
              if (!(j2 < 16 && r == j2 / NG)) upd(r);
            wgam[j * NTR] = gm;
            wmu[j * NTR] = mu;
            wdel[j * NTR] = dl;
            m1o = m1o_n; cA = cA_n; cf = cf_n; ci = ci_n; dj = d_n;
          }
        } else {
        double m1o = Lm1[par][ht], cA = LA[par][ht], dj = LG[par][0];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int jn = (j + 1) & 15, j2 = j + 2;
          double m1o_n = Lm1[par][jn * NTR + ht], cA_n = LA[par][jn * NTR + ht], d_n = LG[par][jn * 33];   // (LA holds c (A + cst))
          const double g_next = LG[par][j * 32 + jn];
          double gl[RPG];                                   // this group's column j of G, for the in-block update below
#pragma unroll
          for (int r = (j / NG); r < RPG; r++) gl[r] = LG[par][j * 32 + hg + NG * r];
          double gxn[CXC ? RPG : 1];                        // (CXC) column j of the NEXT block's X_{b+1}'X_b, this group's rows
          if constexpr (CXC) {
#pragma unroll
            for (int r = 0; r < RPG; r++) gxn[r] = LGx[(b + 1) % 3][(hg + NG * r) * 16 + j];
          }
          __builtin_amdgcn_sched_barrier(0);                // every LDS read that does not depend on this step's delta is on its way
          double s = sb + m1o * dj;                         // cp_Y_X(k,j) - (cp_betaX_X(j,k) - m1*cp_X(j,j))   :71
          double mu = rc_coef * s;                          // :73
          double x = fma(-(s * s), rc_K, cA);               // c*(log(1-Phi) - log Phi - mu^2/(2 sig2) + cst), mu^2 = coef^2 s^2   :75-77
          double gm = aq_sigmoid_neg_fast(x);
          double dl = gm * mu - m1o;                        // m1 - m1_old, m1 = gam*mu   :79
          sb = s_nx - g_next * dl;                          // S of SNP j+1 completed with this step's update: one FMA on the chain
          // in-block part of :81, this group's rows (rows <= j are already consumed: updating them is harmless): first the
          // register that holds SNP j+2, whose S goes on its way to every group at once
          if (j2 < 16) {
            Sown[j2 / NG] -= gl[j2 / NG] * dl;
            s_nx = __shfl(Sown[j2 / NG], (j2 % NG) * NTR + ht, 64);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int r = (j / NG); r < RPG; r++)
            if (!(j2 < 16 && r == j2 / NG)) Sown[r] -= gl[r] * dl;
          if constexpr (CXC) {
#pragma unroll
            for (int r = 0; r < RPG; r++) cxn[r] = (j == 0 ? 0.0 : cxn[r]) + gxn[r] * dl;
          }
          wgam[j * NTR] = gm;
          wmu[j * NTR] = mu;
          wdel[j * NTR] = dl;
          m1o = m1o_n; cA = cA_n; dj = d_n;
          if (j == 7) tl_mark(need, 6);    // (diag) half of the steps
        }
        }
        signal(6, b - seg_b0 + 1);   // delta, gam, mu of block b are in LDS
      };
      if constexpr (NT3 > 0) {
        // chain for block b0 + i - 1, then this wave's share of the matrix work of phase i
        run(std::integral_constant<int, (NT3 > 0 ? NT3 : 1)>{}, std::integral_constant<int, 2>{},
            [&](int i) __attribute__((always_inline)) { if (i >= 1 && i <= nblk) chain_block(seg_b0 + i - 1); });
      } else {
        for (int b = seg_b0; b < seg_b1; b++) chain_block(b);
      }
    } else if constexpr (NT3 > 0) {
      run(std::integral_constant<int, (NT3 > 0 ? NT3 : 1)>{}, std::integral_constant<int, 2>{}, no_hook);   // init mode: R -= X beta on its tiles
    }
    __syncthreads();   // matches the matrix waves' barrier before the final sums
  } else if (helper) {
    // =========================== helper wave ===============================================
    // The transcendental per-entry inputs of the block are computed HERE, a phase ahead, from theta_j + zeta_k (the work
    // of the former p x q pre-pass kernel, which wrote and re-read 16 B per entry):
    //   A = log(1-Phi(u)) - log Phi(u)      src/coreLoop.cpp:75-76 (its log_Phi / log_1_min_Phi inputs, R/...core.R:293-295)
    //   Z = a + gam b,  a = u + imr0/sqrt(c), b = (imr1 - imr0)/sqrt(c) at U = sqrt(c) u       R/update_vb.R:217-234
    // This wave shares its SIMD only with the recurrence wave, whose dependent chain leaves the VALU mostly idle.
    // Instruction priority: at equal priority this wave's fp64 operations alternate with the recurrence wave's 64-cycle MFMAs --
    // one VALU operation per MFMA slot, a crawl; raised, its bursts run at the VALU's own rate and the MFMAs fill the gaps.
    if (a.hprio == 3) __builtin_amdgcn_s_setprio(3);
    else if (a.hprio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.hprio == 1) __builtin_amdgcn_s_setprio(1);
    const size_t tbase = (size_t)(tile0 + (ht >> 4)) * a.p_pad * 16 + (ht & 15);   // this lane's tile and column
    const double zk = a.zeta[ktrait];
    typedef __attribute__((address_space(3))) const double aq_lds_cdouble;
    aq_lds_cdouble *ptab = (aq_lds_cdouble *)(__attribute__((address_space(3))) double *)Lpt;
    const double inv_sqrt_c = 1.0 / a.sqrt_c;
    const double cst_k = MASK ? 0.0 : a.cst[ktrait];                                 // -(log_tau + log_sig2_inv + log sig2_beta) / 2
    double th[RPG];   // theta of the block to be staged next, loaded a phase earlier so that the arithmetic never waits for HBM
    auto theta_load = [&](int b) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RPG; r++) th[r] = a.theta[16 * b + hg + NG * r];
    };
    // complete Y: gam and mu of the block to be staged next, requested together with its theta a phase ahead; stage() turns them
    // into m1 = gam mu first thing, so that their registers are free again while the probit tables are evaluated
    double pg[MASK ? 1 : RPG], pm[MASK ? 1 : RPG];
    auto gm_load = [&](int b) __attribute__((always_inline)) {
      if constexpr (!MASK) {
#pragma unroll
        for (int r = 0; r < RPG; r++) {
          const size_t off = tbase + (size_t)(16 * b + hg + NG * r) * 16;
          pg[r] = a.gam[off];
          pm[r] = a.mu[off];
        }
      }
    };
    // MASK: per-trait constants of the NA forms (src/coreLoop.cpp:108, R/update_vb.R:45) and this block's slice of GK
    const double tau_k = MASK ? a.tau[ktrait] : 1.0;
    const double sig2_inv_h = MASK ? *a.sig2_inv_p : 0.0;
    const double cstna_k = MASK ? -(a.log_tau[ktrait] + *a.log_sig2_inv_p) / 2 : 0.0;
    auto gk_block = [&](int b) __attribute__((always_inline)) { return a.GK + ((size_t)tile0 * a.nb + b) * AQ_GK_STRIDE; };
    // The Gram blocks go global -> LDS by LDS-DMA: no registers, and the whole 50 KB of a block is in flight at once (through
    // registers one wave cannot keep enough loads outstanding: the staging alone took 4x the phase).  hipcc drains vmcnt(0) at
    // the first use of an ordinary load's result while a DMA is pending, so every ordinary load of an iteration is consumed
    // before the DMAs are issued (the helper's own loads run one block ahead, see `pre_load`).
    // Buffers: the diagonal blocks by block parity (read throughout the chain), the cross blocks in a single buffer that the
    // recurrence wave releases (counter Fl[13]) right after the correction at the start of its chain -- the DMA of block b+1 has
    // the whole chain of block b to land.
    auto gk_dma = [&](int b, int par) __attribute__((always_inline)) {
      const char *src = (const char *)gk_block(b) + lane * 16;
#pragma unroll
      for (int i = 0; i < AQ_GK_DIAG / 128; i++) aq_glds16(src + 1024 * i, &LGk2[par][128 * i]);   // 16 lower triangles, 17 KB
      src += AQ_GK_DIAG * 8;
#pragma unroll
      for (int i = 0; i < 32; i++) aq_glds16(src + 1024 * i, &LGxk[128 * i]);                      // 16 cross blocks, 32 KB
    };
    // block b -> LDS parity par.  Complete Y: loads first, the probit arithmetic covers their latency.  MASK: gam, mu and the
    // diagonal X_norm_sq(j,k) of block b were requested by pre_load(b) an iteration earlier.
    double pl_g[MASK ? RPG : 1], pl_m[MASK ? RPG : 1], pl_xn[MASK ? RPG : 1];
    double sv_g[MASK ? RPG : 1], sv_m[MASK ? RPG : 1], sv_xn[MASK ? RPG : 1], sv_th[MASK ? RPG : 1];   // the block being staged
    auto pre_load = [&](int b) __attribute__((always_inline)) {
      const double *gk = gk_block(b);
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        const size_t off = tbase + (size_t)(16 * b + hg + NG * r) * 16;
        const int jj = hg + NG * r;
        pl_g[r] = a.gam[off];
        pl_m[r] = a.mu[off];
        pl_xn[r] = gk[(jj * (jj + 1) / 2 + jj) * 16 + ht];     // X_norm_sq(j,k) = the diagonal of the trait's own block
      }
    };
    // MASK: first half of `stage` -- takes over the pre-loaded values (pinned in registers: nothing of them is waited for
    // once the DMA is pending), starts the DMA of block b's Gram blocks and then requests the values of block b_next, which
    // arrive behind the DMA while block b is computed
    auto stage_dma = [&](int b, int par, int b_next) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        asm volatile("" : "+v"(pl_g[r]), "+v"(pl_m[r]), "+v"(pl_xn[r]), "+v"(th[r]) : : "memory");
        sv_g[r] = pl_g[r]; sv_m[r] = pl_m[r]; sv_xn[r] = pl_xn[r]; sv_th[r] = th[r];
      }
      gk_dma(b, par);
      if (b_next < seg_b1) { pre_load(b_next); theta_load(b_next); }
    };
    auto stage = [&](int b, int par) __attribute__((always_inline)) {
      double st_G[4], st_Gx[4], st_xn[MASK ? RPG : 1];
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < RPG; r++) { Lm1[par][lane + 64 * r] = sv_g[r] * sv_m[r]; st_xn[r] = sv_xn[r]; }
      } else {
#pragma unroll
        for (int r = 0; r < RPG; r++) Lm1[par][lane + 64 * r] = pg[r] * pm[r];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          st_G[r] = a.G[(size_t)b * 256 + lane + 64 * r];
          // the cross Gram block that goes with this staging: block b's own, or (CXC) block b+1's, which the chain of block b reads
          const int bx = CXC ? (b + 1 < a.nb ? b + 1 : b) : b;
          st_Gx[r] = a.Gx[(size_t)bx * 256 + lane + 64 * r];
        }
      }
      // MASK: the per-entry constants of the NA forms -- a reciprocal and a logarithm per entry, each a chain of dependent fp64
      // operations -- for all entries of the lane at once, so that the RPG chains interleave (inside the table passes below every
      // entry sits between scheduling barriers and its chain would run alone; this wave's iteration is on the critical cycle of
      // the MASK instances)
      double ls2r[MASK ? RPG : 1];
      if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < RPG; r++) {
          const int e = lane + 64 * r;
          const double is2 = a.c * (st_xn[r] + sig2_inv_h) * tau_k;            // update_sig2_beta_vb_ with X_norm_sq, R/update_vb.R:45
          const double s2 = aq_recip_pos(is2);
          const double ls2 = -aq_log_pos(is2);
          const double cf = a.c * s2 * tau_k;                                  // src/coreLoop.cpp:125
          Lcoef[par][e] = cf;
          LK[par][e] = cf * cf * (a.c * 0.5 * is2);                            // coef^2 c / (2 sig2_beta), 1 / sig2_beta = is2: x = cA - s^2 K
          Ls2[par][e] = s2;
          Lls2[par][e] = ls2;
          Lxn[par][e] = st_xn[r];
          ls2r[r] = ls2;
        }
      }
      // A(u), the slope b = imr1 - imr0 and the intercept a = u + imr0 of Z (at U = sqrt(c) u, over sqrt(c), when annealing:
      // R/update_vb.R:219-233), u = theta_j + zeta_k, for the RPG entries of this lane.
      auto emitA = [&](int r, double A) __attribute__((always_inline)) {
        const int j = 16 * b + hg + NG * r, e = lane + 64 * r;
        const bool valid = kvalid && j < a.p;
        if constexpr (MASK) {
          LA[par][e] = a.c * ((valid ? A : 0.0) - 0.5 * ls2r[r] + cstna_k);    // src/coreLoop.cpp:127-129
        } else {
          LA[par][e] = a.c * ((valid ? A : 0.0) + cst_k);                        // c (log(1-Phi) - log Phi + cst), src/coreLoop.cpp:75-77
        }
      };
      auto emitZ = [&](int r, double zb, double za) __attribute__((always_inline)) {
        const int j = 16 * b + hg + NG * r, e = lane + 64 * r;
        const bool valid = kvalid && j < a.p;
        LB[par][e] = valid ? zb : 0.0;
        Laa[par][e] = valid ? za : 0.0;
      };
      // the same three probit-dependent fields again, for a run-time r (second pass of a block with lanes outside the tables)
      auto emit2 = [&](int r, double A, double zb, double za) __attribute__((always_inline)) {
        const int j = 16 * b + hg + NG * r, e = lane + 64 * r;
        const bool valid = kvalid && j < a.p;
        LB[par][e] = valid ? zb : 0.0;
        Laa[par][e] = valid ? za : 0.0;
        if constexpr (MASK) LA[par][e] = a.c * ((valid ? A : 0.0) - 0.5 * Lls2[par][e] + cstna_k);
        else LA[par][e] = a.c * ((valid ? A : 0.0) + cst_k);
      };
      double uu[RPG];
      bool inr = true;
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        uu[r] = (MASK ? sv_th[r] : th[r]) + zk;
        inr = inr && fabs(uu[r]) < AQ_PT_R && fabs(a.sqrt_c * uu[r]) < AQ_PT_R;   // (false for NaN)
      }
      {
        // Tables (aq_probit_tab.h: A, b, d as piecewise polynomials of degree AQ_PT_DEG; coefficient k of function f on interval
        // i at Lpt[(f (DEG+1) + k) NI + i]).  The coefficients of an entry are a gather from LDS, and a wave that waits for
        // them can do nothing else: the reads of entry r+1 are issued BEFORE the arithmetic of entry r (two register sets), so
        // that the Horner chains never wait for the LDS round trip.  Two passes -- A at u, then b and d at sqrt(c) u -- keep
        // the register sets at 2 x 22 and 2 x 44 VGPRs.
        constexpr int NC = AQ_PT_DEG + 1, S = AQ_PT_NI, F = NC * S;
        // interval and local variable of v = |x|: i = floor(2 v) (clamped: a lane outside the tables reads the last interval
        // and is redone below), x = 2 (2 v - i) - 1
        auto locate = [&](double x, double *xloc) __attribute__((always_inline)) {
          const double t = fmin(fabs(x) * (1.0 / AQ_PT_W), AQ_PT_NI - 0.5);
          const int i = (int)t;
          *xloc = fma(2.0, t - (double)i, -1.0);
          return ptab + i;
        };
        {
          double co[2][NC], xl[2];
          auto fetch = [&](int r, auto setc) __attribute__((always_inline)) {
            constexpr int set = decltype(setc)::value;
            aq_lds_cdouble *pp = locate(uu[r], &xl[set]);
#pragma unroll
            for (int k = 0; k < NC; k++) co[set][k] = pp[k * S];
          };
          fetch(0, std::integral_constant<int, 0>{});
          aq_static_for<RPG>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value, set = r & 1;
            if constexpr (r + 1 < RPG) fetch(r + 1, std::integral_constant<int, (r + 1) & 1>{});
            __builtin_amdgcn_sched_barrier(0);
            double pa = co[set][NC - 1];
#pragma unroll
            for (int k = NC - 2; k >= 0; k--) pa = fma(pa, xl[set], co[set][k]);
            emitA(r, uu[r] < 0.0 ? -pa : pa);                            // A is odd
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        {
          double co[2][2 * NC], xl[2];
          auto fetch = [&](int r, auto setc) __attribute__((always_inline)) {
            constexpr int set = decltype(setc)::value;
            aq_lds_cdouble *pp = locate(a.sqrt_c * uu[r], &xl[set]);
#pragma unroll
            for (int k = 0; k < NC; k++) {
              co[set][k] = pp[F + k * S];
              co[set][NC + k] = pp[2 * F + k * S];
            }
          };
          fetch(0, std::integral_constant<int, 0>{});
          aq_static_for<RPG>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value, set = r & 1;
            if constexpr (r + 1 < RPG) fetch(r + 1, std::integral_constant<int, (r + 1) & 1>{});
            __builtin_amdgcn_sched_barrier(0);
            double pb = co[set][NC - 1], pd = co[set][2 * NC - 1];
#pragma unroll
            for (int k = NC - 2; k >= 0; k--) {
              pb = fma(pb, xl[set], co[set][k]);
              pd = fma(pd, xl[set], co[set][NC + k]);
            }
            const double d = uu[r] < 0.0 ? -pd : pd;                     // d is odd, b is even
            // imr0 = -M(U) = -(b + d) / 2:  a = u + imr0 / sqrt(c),  slope = b / sqrt(c)
            emitZ(r, pb * inv_sqrt_c, uu[r] - 0.5 * inv_sqrt_c * (pb + d));
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      }
      if (__builtin_expect(!__all(inr), 0)) {
        // some lane of the wave is outside the tables (|u| >= 12): the block is done again entry by entry with the tail series
        // of aq_special.h for those lanes (aq_probit_tab_eval picks per lane)
#pragma unroll 1
        for (int r = 0; r < RPG; r++) {
          double ur = uu[0];
#pragma unroll
          for (int k = 1; k < RPG; k++) ur = (r == k) ? uu[k] : ur;       // (no dynamically indexed register arrays)
          double A, bM, dM, dummy;
          aq_probit_tab_eval<1>(ur, ptab, &A, &dummy, &dummy);
          aq_probit_tab_eval<2>(a.sqrt_c * ur, ptab, &dummy, &bM, &dM);
          emit2(r, A, bM * inv_sqrt_c, ur - 0.5 * inv_sqrt_c * (bM + dM));
        }
      }
      if constexpr (MASK) {
#ifdef AQ_DIAG_TIME
        const long long t_dma = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the Gram blocks have landed
#ifdef AQ_DIAG_TIME
        dg_wait += __builtin_readcyclecounter() - t_dma;   // (shows up as the helper's "waiting on matrix/recurrence")
#endif
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int e = lane + 64 * r;
          LG[par][(e >> 4) * 32 + (e & 15)] = st_G[r];
          LGx[CXC ? (b + 1) % 3 : par][e] = st_Gx[r];
        }
      }
    };
    // running column sums of this lane's entries (they all belong to trait ht): sum gam, sum m2, sum beta^2 (MASK: sum
    // X_norm_sq (m2 - beta^2), R/update_vb.R:152-154), sum Z, and for MASK sum gam log sig2_beta
    double cs0 = 0.0, cs1 = 0.0, cs2 = 0.0, cs3 = 0.0, cs5 = 0.0;
    auto finalize = [&](int b, int par) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        const int e = lane + 64 * r, hj = hg + NG * r;
        const double gm = Lgam[par][e], mu = Lmu[par][e];
        const size_t off = tbase + (size_t)(16 * b + hj) * 16;
        if (lead) {
          a.gam[off] = gm;
          a.mu[off] = mu;
        }
        const int j = 16 * b + hj;
        double gb = 0.0;
        if (kvalid && j < a.p) {
          const double be = gm * mu;
          gb = Laa[par][e] + gm * LB[par][e];       // Z_jk = a + gam b: its row and column sums are all that is needed
          cs0 += gm;
          cs3 += gb;
          if constexpr (MASK) {
            const double s2 = Ls2[par][e];
            const double m2 = (mu * mu + s2) * gm;                           // update_m2_beta_, R/update_vb.R:19-31
            const double xn = Lxn[par][e];
            cs1 += m2;
            cs2 += xn * (m2 - be * be);
            cs5 += gm * Lls2[par][e];
          } else {
            cs1 += (mu * mu + sig2b_k) * gm;        // update_m2_beta_, R/update_vb.R:19-31
            cs2 += be * be;
          }
        }
        gb = aq_row16_sum(gb);                      // over the 16 traits of the tile (one DPP row)
        if ((ht & 15) == 0 && lead) a.rowGB[(size_t)(tile0 + (ht >> 4)) * a.p_pad + j] = gb;
      }
    };
    // sample split, exchange on this wave (a.xhelper): a block ahead of the chain -- S' of block b is complete a phase before the
    // chain of block b starts -- so the round trip through the shared cache level is off the recurrence wave's path
    auto exchange = [&](int b) __attribute__((always_inline)) {
      const int par = b & 1, kdone = b - seg_b0 + 1;
#pragma unroll
      for (int m = 0; m < NWM; m++) wait_ge(m, kdone);
      if constexpr (NT3 > 0) wait_ge(11, kdone);   // the recurrence wave's own tiles
      double own[RPG];
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        const int e = lane + 64 * r;
        double sv = Sp[par][0][e];
#pragma unroll
        for (int ww = 1; ww < NPS; ww++) sv += Sp[par][ww][e];
        own[r] = sv;
      }
      split_exchange(b, own, [] {});
#pragma unroll
      for (int r = 0; r < RPG; r++) Stot[par][lane + 64 * r] = own[r];
      signal(14, kdone);
    };
    if (a.mode == 1) {
      // init mode: R = Y - X beta.  This wave plays the recurrence's part: beta of block k goes where delta would (Ldel, by block
      // parity) and is announced on the recurrence counter; the matrix waves apply it in their phase k + 2, whose end they
      // announce as k + 3 -- so the buffer of block k - 2 is free once every matrix wave has announced k + 1.
      for (int b = seg_b0; b < seg_b1; b++) {
        const int k = b - seg_b0, par = b & 1;
        if (k >= 2) {
#pragma unroll
          for (int m = 0; m < NWM; m++) wait_ge(m, k + 1);
          if (NT3 > 0) wait_ge(11, k + 1);
        }
#pragma unroll
        for (int r = 0; r < RPG; r++) {
          const int e = lane + 64 * r, hj = hg + NG * r;
          const size_t off = tbase + (size_t)(16 * b + hj) * 16;
          double gm = a.gam[off], mu = a.mu[off];
          double be = gm * mu;                                  // update_beta_vb_, R/update_vb.R:17
          Ldel[par][e] = be;
          if (kvalid && (16 * b + hj) < a.p) {
            const double m2 = (mu * mu + sig2b_k) * gm;         // initial m2_beta (q-vector sig2_beta_vb), R/atlasqtl_global_local_core.R:113
            cs0 += gm;
            cs1 += m2;
            if constexpr (MASK) cs2 += gk_block(b)[(hj * (hj + 1) / 2 + hj) * 16 + ht] * (m2 - be * be);
            else cs2 += be * be;
          }
        }
        signal(6, k + 1);
      }
    } else {
      // L2 warm-up of the X operand panels: the workgroups of an XCD walk through the SNP blocks nearly in step, and whichever
      // matrix wave touches a panel first waits for HBM (its operand prefetch reaches only a tile and a half ahead).  Every
      // helper wave therefore touches 8 KB slices (its lane: one 128-B line each) of the panels the NEXT phase will read -- the
      // S' operands of block b+2 and the update operands of block b -- a phase ahead; the workgroups that share an XCD (block
      // index modulo 8 under the round-robin placement: speed only) take different slices, up to four each.  The loaded words
      // are never used; their registers stay reserved until the wait at the next call.  C3: 34.96 -> 34.49 ms.
      int xt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const long long xt_blk = (long long)NTT * C * 2048;                      // bytes of one panel
      const int xt_nsl = (int)((xt_blk + 8191) / 8192);                        // slices of a panel
      const int xt_wg = (int)(gridDim.x < 256 ? gridDim.x : 256) / 8 > 0 ? (int)(gridDim.x < 256 ? gridDim.x : 256) / 8 : 1;   // workgroups of this launch per XCD
      auto x_touch = [&](int bA, int bU) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xt[0]), "+v"(xt[1]), "+v"(xt[2]), "+v"(xt[3]), "+v"(xt[4]), "+v"(xt[5]), "+v"(xt[6]), "+v"(xt[7]));
        const char *pa = (const char *)a.XA + (long long)bA * xt_blk + lane * 128, *pu = (const char *)a.XU + (long long)bU * xt_blk + lane * 128;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sl = (int)((blockIdx.x >> 3) % xt_wg) + k * xt_wg;
          if (sl < xt_nsl && (long long)sl * 8192 + lane * 128 < xt_blk) {
            asm volatile("global_load_dword %0, %1, off" : "=v"(xt[2 * k]) : "v"(pa + (long long)sl * 8192));
            asm volatile("global_load_dword %0, %1, off" : "=v"(xt[2 * k + 1]) : "v"(pu + (long long)sl * 8192));
          }
        }
      };
      theta_load(seg_b0);
      gm_load(seg_b0);
      if constexpr (MASK) { pre_load(seg_b0); stage_dma(seg_b0, seg_b0 & 1, seg_b0 + 1); }
      stage(seg_b0, seg_b0 & 1);
      signal(7, 1);
      if constexpr (!SEG && TT == 1) { if (C > 1 && a.xhelper) exchange(seg_b0); }
      if constexpr (!MASK) { if (seg_b0 + 1 < seg_b1) { theta_load(seg_b0 + 1); gm_load(seg_b0 + 1); } }
      for (int b = seg_b0; b < seg_b1; b++) {
        const int par = b & 1;
        if constexpr (MASK) {
          // the Gram blocks of block b+1 go on their way as soon as the chain of block b has applied the cross blocks (which
          // also means that block b-1, the last reader of the other diagonal buffer, is through)
          if (b + 1 < seg_b1) { wait_ge(13, b - seg_b0 + 1); stage_dma(b + 1, par ^ 1, b + 2); }
        }
        // block b-1 must be through the recurrence: its gam / mu are read here, and the parity buffers about to be
        // overwritten with block b+1 are the ones it read
        if (b > seg_b0) {
          wait_ge(6, b - seg_b0);
          if constexpr (!MASK && !CXC) {
            // cross-block correction X_b'X_{b-1} delta_{b-1} of block b for the recurrence wave (a 16 x 16 by 16 x NTR product: 128
            // multiply-adds and 144 LDS reads per lane that would otherwise sit between "S' complete" and the first chain step)
            double dlp[16];
#pragma unroll
            for (int i = 0; i < 16; i++) dlp[i] = Ldel[par ^ 1][i * NTR + ht];
#pragma unroll
            for (int r = 0; r < RPG; r++) {
              double cx = 0.0;
              const double *gx = &LGx[par][(hg + NG * r) * 16];
#pragma unroll
              for (int i = 0; i < 16; i++) cx += gx[i] * dlp[i];
              Lcx[lane + 64 * r] = cx;
            }
            signal(12, b - seg_b0 + 1);
          }
          finalize(b - 1, par ^ 1);
        }
        if (b + 1 < seg_b1) {
          stage(b + 1, par ^ 1);
          signal(7, b - seg_b0 + 2);
          if constexpr (!SEG && TT == 1) { if (C > 1 && a.xhelper) exchange(b + 1); }
          if (a.xtouch && b + 2 < seg_b1) x_touch(b + 2, b);
          if constexpr (!MASK) { if (b + 2 < seg_b1) { theta_load(b + 2); gm_load(b + 2); } }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xt[0]), "+v"(xt[1]), "+v"(xt[2]), "+v"(xt[3]), "+v"(xt[4]), "+v"(xt[5]), "+v"(xt[6]), "+v"(xt[7]));
      wait_ge(6, nblk);
      finalize(seg_b1 - 1, (seg_b1 - 1) & 1);
    }
    Lred[0][lane] = cs0; Lred[1][lane] = cs1; Lred[2][lane] = cs2; Lred[3][lane] = cs3; Lred[5][lane] = cs5;   // [row group][trait]
    __syncthreads();   // matches the matrix waves' barrier before the final sums
  } else {
    // =========================== matrix waves ==============================================
    if (mw < 3) run(std::integral_constant<int, NT>{}, std::integral_constant<int, 0>{}, no_hook);
    else run(std::integral_constant<int, NT2>{}, std::integral_constant<int, 1>{}, no_hook);
    __syncthreads();
  }
#ifdef AQ_DIAG_TIME
  if (lane == 0 && a.dbg) {
    a.dbg[((size_t)blockIdx.x * 8 + w) * 3] = dg_wait;
    a.dbg[((size_t)blockIdx.x * 8 + w) * 3 + 1] = dg_wait_b;
    a.dbg[((size_t)blockIdx.x * 8 + w) * 3 + 2] = __builtin_readcyclecounter() - dg_t0;
  }
#endif
  // ---- per-trait sums ----
  if (tid < NTR) {
    int k2 = tile0 * 16 + tid;
    double r2 = 0.0;
    for (int s = 0; s < NPS * 4; s++) r2 += Lrn[s][tid];
    double *sm = a.sums + (size_t)seg * (MASK ? 6 : 5) * a.q_pad;   // MASK: the six rows of the NA forms (aq_core_sweep_mis.h)
    if (C > 1) a.rnpart[(size_t)part * a.q_pad + k2] = r2;   // added over the parts by aq_k_sum_parts
    else sm[(size_t)4 * a.q_pad + k2] = r2;
    for (int v = 0; v < 6 && lead; v++) {
      if (v == 4 || (v == 5 && !MASK)) continue;
      double acc2 = 0.0;
      for (int jj = 0; jj < NG; jj++) acc2 += Lred[v][jj * NTR + tid];
      sm[(size_t)v * a.q_pad + k2] = acc2;
    }
  }
  if (SEG) {
    // publish this group's residual: every storing wave drains its stores, then one agent-scope release + flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&a.done[wg], seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
