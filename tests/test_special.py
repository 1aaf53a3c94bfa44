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


def test_log_ndtr_pair_one_erfc():
    x = np.concatenate([np.linspace(-60, 60, 4001), [-37.0, 37.0, 0.0, 1e-300, -1e-300]])
    for which, ref in ((5, sp.log_ndtr(x)), (6, sp.log_ndtr(-x))):
        got = ev(which, x)
        # (erfc's far tail, where log Phi ~ -1e-262, is good to ~6e-14 relative in glibc)
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)) < 2e-13


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
