"""fp64 special functions of the HIP path (atlasqtl_amd/csrc/aq_special.h), evaluated on the host
through the C ABI test hooks, against SciPy (the oracle's provider of the same functions)."""
import ctypes as C

import numpy as np
from scipy import special as sp

from atlasqtl_amd import _lib
from oracle import atlasqtl_oracle as O


def ev(which, x, x2=None):
    L = _lib.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    a2 = np.ascontiguousarray(x2, dtype=np.float64) if x2 is not None else None
    rc = L.aq_special_eval(which, _lib.as_dp(x), _lib.as_dp(a2) if a2 is not None else None, _lib.as_dp(out), x.size)
    assert rc == 0
    return out


def test_log_ndtr_both_tails():
    x = np.concatenate([np.linspace(-60, 12, 2001), [-37.0, -36.999, 0.0, 1e-300, -1e-300, 38.0]])
    got, ref = ev(0, x), sp.log_ndtr(x)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) < 5e-14


def _mp_log_ndtr(v):
    import mpmath as mp
    mp.mp.dps = 60
    v = mp.mpf(float(v))
    if v > 0:
        return mp.log1p(-mp.erfc(v / mp.sqrt(2)) / 2)
    return mp.log(mp.erfc(-v / mp.sqrt(2)) / 2)


def test_probit_terms_against_mpmath():
    """aq_probit_terms (one erfcx, one exp, one log, one log1p): both log tails and both inverse Mills ratios
    (pnorm(log.p = TRUE) and inv_mills_ratio_, R/utils.R:172-191) against 60-digit mpmath; SciPy's own log_ndtr is
    only good to 2e-13 in the far tails, so it is not the yardstick here."""
    import mpmath as mp
    x = np.concatenate([np.linspace(-40, 40, 801), np.linspace(-60, 60, 121), [-37.0, 37.0, 0.0, 1e-300, -1e-300, 100.0]])
    ref = np.array([float(_mp_log_ndtr(v)) for v in x])
    relerr = lambda got, want: np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))
    assert relerr(ev(5, x), ref) < 4e-15
    assert relerr(ev(6, -x), ref) < 4e-15
    assert relerr(ev(5, x), sp.log_ndtr(x)) < 5e-13
    mp.mp.dps = 60
    phi = lambda v: mp.exp(-mp.mpf(float(v)) ** 2 / 2) / mp.sqrt(2 * mp.pi)
    imr1 = np.array([float(phi(v) / (mp.erfc(-mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    imr0 = np.array([float(-phi(v) / (mp.erfc(mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    assert relerr(ev(7, x), imr1) < 4e-15
    assert relerr(ev(8, x), imr0) < 4e-15
    # the pre-pass form (one log, one reciprocal): A = log(1-Phi) - log Phi carries an absolute error of a few ulp of 1
    A_ref = np.array([float(_mp_log_ndtr(-v) - _mp_log_ndtr(v)) for v in x])
    assert np.max(np.abs(ev(10, x) - A_ref) / np.maximum(np.abs(A_ref), 1.0)) < 4e-15
    assert relerr(ev(11, x), imr1) < 6e-15
    assert relerr(ev(12, x), imr0) < 6e-15
    z = np.concatenate([np.linspace(0, 30, 301), [1e-300, 1e3, 1e8]])
    mp.mp.dps = 60

    def erfcx(v):
        v = mp.mpf(float(v))
        if v < 25:
            return mp.exp(v * v) * mp.erfc(v)
        s, term = mp.mpf(1), mp.mpf(1)
        for m in range(1, 40):
            term *= -(2 * m - 1) / (2 * v * v)
            s += term
        return s / (v * mp.sqrt(mp.pi))
    assert relerr(ev(9, z), np.array([float(erfcx(v)) for v in z])) < 2e-15


def test_digamma():
    x = np.concatenate([np.geomspace(1e-3, 1e6, 500), [0.5, 1.0, 1.01, 9.99, 10.0, 505.0]])
    assert np.max(np.abs(ev(1, x) - sp.digamma(x)) / np.maximum(np.abs(sp.digamma(x)), 1e-3)) < 1e-13


def test_digamma_never_spins_on_bad_input():
    out = ev(1, np.array([-1e300, -3.0, 0.0, np.nan]))
    assert np.all(np.isnan(out))


def test_expint_e1_small():
    x = np.geomspace(1e-12, 1.0, 400)
    assert np.max(np.abs(ev(2, x) - sp.exp1(x)) / sp.exp1(x)) < 5e-15


def test_gamma_inc_upper():
    rng = np.random.default_rng(0)
    a = rng.uniform(0.005, 1.5, size=4000)
    x = np.geomspace(1e-10, 200.0, 4000)
    ref = sp.gamma(a) * sp.gammaincc(a, x)
    got = ev(3, x, a)
    ok = ref > 1e-280
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 2e-13


def test_sigmoid_matches_reference_form():
    x = np.linspace(-745, 745, 5001)
    ref = np.exp(-O.log_one_plus_exp_(x))          # src/coreLoop.cpp:28-33,75
    got = ev(4, x)
    ok = ref > 1e-300
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 1e-13


def test_short_chain_sigmoid():
    """aq_sigmoid_neg_fast (Estrin exp + reciprocal in [1,2]) against the reference form exp(-log(1+exp(x))),
    src/coreLoop.cpp:28-33,75-77, in 60-digit arithmetic."""
    import mpmath as mp
    mp.mp.dps = 60
    x = np.concatenate([np.linspace(-800, 800, 1601), np.linspace(-40, 40, 4001), [0.0, 1e-300, -1e-300, 1e4, -1e4, 744.9, 745.5]])
    ref = np.array([float(1 / (1 + mp.exp(mp.mpf(float(v))))) for v in x])
    got = ev(13, x)
    assert np.max(np.abs(got - ref) / np.maximum(ref, 1e-300)) < 4e-16 * 4
    assert np.max(np.abs(ev(4, x) - ref) / np.maximum(ref, 1e-300)) < 4e-16 * 4


def test_q_approx_vec_shared_stopping_rule():
    """exp(x) E1(x) with the reference's SHARED Lentz iteration count (R/utils.R:380-423, note N2)."""
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for x in (rng.uniform(1e-6, 50, 300), np.array([0.3, 0.9]), np.array([1.0000001, 3.0, 1e4]), np.array([2.5])):
        ref, iters_ref = O.Q_approx_vec(x, return_iters=True)
        out = np.empty_like(x)
        it = C.c_int32(0)
        assert L.aq_q_approx_vec(_lib.as_dp(np.ascontiguousarray(x)), _lib.as_dp(out), x.size, C.byref(it)) == 0
        assert it.value == iters_ref
        assert np.max(np.abs(out - ref) / ref) < 1e-13


def test_hs_integral_matches_the_oracle_restatement():
    """aq_hs_integral (compute_integral_hs_, R/utils.R:425-568, the four (m, n) of df = 5 and 7; term lists written out by hand
    for the device) against the oracle's restatement with the reference's generic loops, where the lists are well conditioned."""
    x = 10 ** np.random.default_rng(5).uniform(-3, 0.7, 300)                  # L = beta / alpha
    Q = np.exp(x) * sp.exp1(x)
    for which, (df, m, n) in {14: (5, 3, 3), 15: (5, 3, 2), 16: (7, 4, 4), 17: (7, 4, 3)}.items():
        ref = np.array([O.compute_integral_hs_(float(df), xi * df, m, n, qi) for xi, qi in zip(x, Q)])
        assert np.max(np.abs(ev(which, x, Q) - ref) / np.abs(ref)) < 1e-11, which


def _probit_refs(x):
    import mpmath as mp
    mp.mp.dps = 60
    phi = lambda v: mp.exp(-mp.mpf(float(v)) ** 2 / 2) / mp.sqrt(2 * mp.pi)
    A = np.array([float(_mp_log_ndtr(-v) - _mp_log_ndtr(v)) for v in x])
    imr1 = np.array([float(phi(v) / (mp.erfc(-mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    imr0 = np.array([float(-phi(v) / (mp.erfc(mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    return A, imr1, imr0


def probit_table_points():
    rng = np.random.default_rng(7)
    edges = np.arange(0, 12.5, 0.5)                                   # interval boundaries and their neighbours
    near = np.concatenate([edges, np.nextafter(edges, -1), np.nextafter(edges, 100)])
    x = np.concatenate([rng.uniform(-12, 12, 6000), near, -near, np.linspace(-13, 13, 261), [0.0, -0.0, 1e-300, -1e-300, 11.999999,
                        12.0, -12.0, 12.000001, 40.0, -40.0, 1e4]])
    return x


def test_probit_tables_against_mpmath():
    """The table-driven probit terms of the sweep kernel (aq_probit_tab.h, piecewise degree-10 polynomials of A, b = imr1 - imr0
    and d; closed forms beyond |x| = 12) against 60-digit arithmetic: A to a few ulp of max(|A|, 1), the Mills ratios to a few
    ulp of the LARGER of the two (they enter Z = a + gam b only through b = imr1 - imr0 and a = x + imr0, R/update_vb.R:217-234),
    and the reference's clamps (R/utils.R:180-181,188-189) hold."""
    x = probit_table_points()
    A, imr1, imr0 = _probit_refs(x)
    gA, g1, g0 = ev(18, x), ev(19, x), ev(20, x)
    assert np.max(np.abs(gA - A) / np.maximum(np.abs(A), 1.0)) < 4e-15
    scale = np.maximum(np.maximum(np.abs(imr1), np.abs(imr0)), 1.0)
    assert np.max(np.abs(g1 - imr1) / scale) < 4e-15
    assert np.max(np.abs(g0 - imr0) / scale) < 4e-15
    assert np.max(np.abs((g1 - g0) - (imr1 - imr0)) / scale) < 4e-15          # the slope b
    # the intercept a = x + imr0 cancels for x >> 1 (a ~ -1/x): a few ulp of x is all fp64 can give, here as before
    assert np.max(np.abs((x + g0) - (x + imr0)) / np.maximum(np.abs(x), 1.0)) < 2e-15
    assert np.all(g1 >= -x) and np.all(g0 <= -x)
    # continuity across the interval boundaries and at the hand-over to the closed forms
    for e in np.arange(0.5, 12.5, 0.5):
        lo, hi = np.nextafter(e, -1), e
        for w in (18, 19, 20):
            a, b = ev(w, np.array([lo, hi, -lo, -hi]))[[0, 1]], ev(w, np.array([lo, hi, -lo, -hi]))[[2, 3]]
            assert abs(a[0] - a[1]) < 1e-13 * max(1.0, abs(a[0])) and abs(b[0] - b[1]) < 1e-13 * max(1.0, abs(b[0]))
    assert np.all(np.isnan(ev(18, np.array([np.nan]))))


def _log_ndtr_pair_refs(x):
    import mpmath as mp
    mp.mp.dps = 60
    lP = np.array([float(mp.log(mp.erfc(-mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    l1 = np.array([float(mp.log(mp.erfc(mp.mpf(float(v)) / mp.sqrt(2)) / 2)) for v in x])
    return lP, l1


def test_log_ndtr_pair_tables_against_mpmath():
    """log Phi(x) and log(1 - Phi(x)) as the ELBO pass takes them (aq_log_ndtr_pair_tab: the A table plus N(v) = log Phi(v),
    tail series beyond |x| = 12; R/elbo.R:10-34 reads pnorm(., log.p = TRUE) of both tails) against 60-digit arithmetic: a few
    ulp of max(|value|, 1) -- the near side is O(1e-33) beyond the tables and comes out as 0."""
    x = probit_table_points()
    lP, l1 = _log_ndtr_pair_refs(x)
    gP, g1 = ev(24, x), ev(25, x)
    assert np.max(np.abs(gP - lP) / np.maximum(np.abs(lP), 1.0)) < 4e-15
    assert np.max(np.abs(g1 - l1) / np.maximum(np.abs(l1), 1.0)) < 4e-15
    # the same two numbers from the closed forms the other kernels use
    assert np.max(np.abs(gP - ev(5, x)) / np.maximum(np.abs(lP), 1.0)) < 4e-15
    assert np.max(np.abs(g1 - ev(6, x)) / np.maximum(np.abs(l1), 1.0)) < 4e-15
    assert np.all(np.isnan(ev(24, np.array([np.nan])))) and np.all(np.isnan(ev(25, np.array([np.nan]))))


def test_annealed_lam2_inv_df_gt_1_against_tricomi_u():
    """update_annealed_lam2_inv_vb_ for df = 3, 5, 7 (R/update_vb.R:76-81).  In Tricomi's U the reference's quotient is
    a U(a + 1, 3 - c, L) / (df U(a, 2 - c, L)), a = c (df - 1) / 2 + 1.  The reference writes each U as a DIFFERENCE of two Kummer
    functions that grow like e^L while the difference falls like L^-a, so its own expression loses digits as L grows -- measured
    here for the oracle's restatement (scipy.special.hyp1f1) and for the device's (power series) against 60-digit mpmath.hyperu:
    ~1e-11 up to L = 2 (where the runs live: L is of order one during the ladder), ~1e-7 up to 6, nothing left beyond 10.
    The device reproduces the expression term by term, as everywhere: the bars are those of the expression, not tighter."""
    import mpmath as mp
    mp.mp.dps = 60
    rng = np.random.default_rng(3)
    for which, df in ((21, 3), (22, 5), (23, 7)):
        L = np.concatenate([rng.uniform(0.01, 2.0, 300), rng.uniform(2.0, 6.0, 100)])
        c = rng.uniform(0.5, 0.93, L.size)                     # the ladder of anneal = c(1, 2, 10) ends at 2^(-1/9) = 0.926
        got = ev(which, L, c)
        ref = np.array([float(mp.mpf(ci) * (df - 1) / 2 + 1) * float(mp.hyperu(ci * (df - 1) / 2 + 2, 3 - ci, li) /
                                                                    mp.hyperu(ci * (df - 1) / 2 + 1, 2 - ci, li)) / df
                        for li, ci in zip(L.tolist(), c.tolist())])
        orc = np.array([O.update_annealed_lam2_inv_vb_(np.array([li]), ci, df)[0] for li, ci in zip(L, c)])
        for lo, hi, tol in ((0, 2, 1e-9), (2, 6, 1e-5)):
            m = (L >= lo) & (L < hi)
            assert np.max(np.abs(got[m] - ref[m]) / np.abs(ref[m])) < tol, (df, lo)
            assert np.max(np.abs(got[m] - orc[m]) / np.abs(orc[m])) < tol, (df, lo)
        assert np.all(np.isfinite(ev(which, np.array([20.0, 100.0, 600.0]), np.array([0.7, 0.7, 0.7]))))
