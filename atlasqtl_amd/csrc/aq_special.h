// aq_special.h -- fp64 special functions used on the VB path, written once for
// device (hipcc) and host (g++, for the CPU unit tests in tests/test_special.py).
//
// Reference call sites these replace (the reference gets them from base R / GSL,
// whose sources are not part of the reference repository):
//   aq_log_ndtr      pnorm(x, log.p = TRUE)                R/atlasqtl_global_local_core.R:62-63,294-295, R/update_vb.R:223-224
//   aq_digamma       digamma()                             R/update_vb.R:120,159
//   aq_expint_e1     gsl::expint_E1 (x <= 1 branch)        R/utils.R:387
//   aq_lentz_*       the modified-Lentz loop for x > 1     R/utils.R:392-419
//   aq_gamma_inc_upper  gsl::gamma_inc(a, x), a in (0, 2)  R/update_vb.R:74
//   aq_sigmoid_neg   exp(-logOnePlusExp(x))                src/coreLoop.cpp:28-33,75-77
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AQ_HD __host__ __device__ __forceinline__
#else
#define AQ_HD static inline
#endif

#define AQ_LOG_SQRT_2PI 0.91893853320467274178032973640562
#define AQ_INV_SQRT2 0.70710678118654752440084436210485
#define AQ_EULER_GAMMA 0.57721566490153286060651209008240

// log Phi(x).  x > 0: log1p(-erfc(x/sqrt2)/2); -37 < x <= 0: log(erfc(-x/sqrt2)/2)
// (erfc keeps full relative accuracy in its tail until it underflows near 26.5);
// x <= -37: asymptotic Mills-ratio series.
AQ_HD double aq_log_ndtr(double x) {
  if (x > 0.0) return log1p(-0.5 * erfc(x * AQ_INV_SQRT2));
  if (x > -37.0) return log(0.5 * erfc(-x * AQ_INV_SQRT2));
  double ix2 = 1.0 / (x * x);
  // 1 - 1/x^2 + 3/x^4 - 15/x^6 + 105/x^8 - 945/x^10
  double ser = 1.0 + ix2 * (-1.0 + ix2 * (3.0 + ix2 * (-15.0 + ix2 * (105.0 - 945.0 * ix2))));
  return -0.5 * x * x - log(-x) - AQ_LOG_SQRT_2PI + log(ser);
}

// log Phi(x) and log(1 - Phi(x)) = log Phi(-x) from ONE erfc: with e = erfc(|x|/sqrt2)/2 (the tail on the far
// side of x), the near-side value is log1p(-e) and the far-side one log(e).  Same formulas as aq_log_ndtr.
AQ_HD void aq_log_ndtr_pair(double x, double *lP, double *l1) {
  double ax = fabs(x);
  double near_, far_;
  if (ax < 37.0) {
    double e = 0.5 * erfc(ax * AQ_INV_SQRT2);
    near_ = log1p(-e);
    far_ = log(e);
  } else {
    near_ = aq_log_ndtr(ax);
    far_ = aq_log_ndtr(-ax);
  }
  if (x > 0.0) { *lP = near_; *l1 = far_; } else { *lP = far_; *l1 = near_; }
}

// exp(-log(1+exp(x))) evaluated as the reference's logOnePlusExp does its case
// split (m = max(x,0)): x < 0 -> 1/(1+e^x); x >= 0 -> e^-x/(1+e^-x).
AQ_HD double aq_sigmoid_neg(double x) {
  double e = exp(-fabs(x));
  double num = (x < 0.0) ? 1.0 : e;
  return num / (1.0 + e);
}

// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series.
AQ_HD double aq_digamma(double x) {
  if (!(x > 0.0)) return NAN;          // the path only calls it with positive arguments; never loop on garbage
  double acc = 0.0;
  for (int i = 0; i < 10 && x < 10.0; i++) {   // at most 10 steps for x > 0
    acc -= 1.0 / x;
    x += 1.0;
  }
  double ix = 1.0 / x, ix2 = ix * ix;
  // B2/2=1/12, B4/4=-1/120, B6/6=1/252, B8/8=-1/240, B10/10=1/132, B12/12=-691/32760, B14/14=1/12
  double ser = ix2 * (1.0 / 12.0 - ix2 * (1.0 / 120.0 - ix2 * (1.0 / 252.0 - ix2 * (1.0 / 240.0 - ix2 * (1.0 / 132.0
               - ix2 * (691.0 / 32760.0 - ix2 * (1.0 / 12.0)))))));
  return acc + log(x) - 0.5 * ix - ser;
}

// E1(x) for 0 < x <= 1: -gamma - ln x - sum_{k>=1} (-x)^k / (k k!)
AQ_HD double aq_expint_e1_small(double x) {
  double sum = 0.0, term = 1.0;
  for (int k = 1; k <= 30; k++) {
    term *= -x / (double)k;
    sum += term / (double)k;
    if (fabs(term) < 1e-18 * fabs(sum)) break;
  }
  return -AQ_EULER_GAMMA - log(x) - sum;
}

// One element's modified-Lentz state for exp(x) E1(x), x > 1, exactly as the
// reference iterates it (R/utils.R:392-419): eps1 = 1e-30.  step(j) performs the
// body for iteration counter j (the reference increments j first, so the first
// call is j = 2) and returns |Delta - 1|.
struct AqLentz {
  double f, C, D;
};
AQ_HD void aq_lentz_init(AqLentz *s) {
  s->f = 1e-30;
  s->C = 1e-30;
  s->D = 0.0;
}
AQ_HD double aq_lentz_step(AqLentz *s, double x, int j) {
  double jm1sq = (double)(j - 1) * (double)(j - 1);
  double Dc = x + 2.0 * j - 1.0 - jm1sq * s->D;
  double Cc = x + 2.0 * j - 1.0 - jm1sq / s->C;
  Dc = 1.0 / Dc;
  double Delta = Cc * Dc;
  s->f = s->f * Delta;
  s->C = Cc;
  s->D = Dc;
  return fabs(Delta - 1.0);
}
AQ_HD double aq_lentz_finish(const AqLentz *s, double x) { return 1.0 / (x + 1.0 + s->f); }

// Unnormalised upper incomplete gamma Gamma(a, x) for 0 < a < 2, x > 0.
//   x >= 1 : modified-Lentz continued fraction (Numerical-Recipes form), relative 1e-16
//   x <  1 : Gamma(a,1) + int_x^1 t^(a-1) e^-t dt, the integral as the alternating
//            series sum_k (-1)^k/k! (1 - x^(a+k))/(a+k), first term via expm1 so a -> 0 is safe
AQ_HD double aq_gamma_inc_cf(double a, double x) {
  const double tiny = 1e-300;
  double b = x + 1.0 - a;
  double c = 1.0 / tiny;
  double d = 1.0 / b;
  double h = d;
  for (int i = 1; i <= 500; i++) {
    double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return exp(-x + a * log(x)) * h;
}
AQ_HD double aq_gamma_inc_upper(double a, double x) {
  if (x >= 1.0) return aq_gamma_inc_cf(a, x);
  double g1 = aq_gamma_inc_cf(a, 1.0);
  double lx = log(x);
  double sum = 0.0, sgn_over_fact = 1.0;
  for (int k = 0; k <= 40; k++) {
    double ak = a + (double)k;
    double t = sgn_over_fact * (-expm1(ak * lx)) / ak;
    sum += t;
    if (k > 0 && fabs(t) < 1e-18 * fabs(sum)) break;
    sgn_over_fact *= -1.0 / (double)(k + 1);
  }
  return g1 + sum;
}

// Inverse Mills ratios and the probit auxiliary mean Z of one (j,k) entry,
// R/update_vb.R:217-234 with R/utils.R:172-191.  U = sqrt_c*u, lP = log Phi(U), l1 = log(1-Phi(U)).
AQ_HD double aq_probit_z(double gam, double u, double U, double lP, double l1, double sqrt_c) {
  double base = -0.5 * U * U - AQ_LOG_SQRT_2PI;
  double imr1 = exp(base - lP);
  if (imr1 < -U) imr1 = -U;
  double imr0 = -exp(base - l1);
  if (imr0 > -U) imr0 = -U;
  return (gam * (imr1 - imr0) + imr0) / sqrt_c + u;
}
