"""GPU: the device build of the fp64 special functions (ocml exp/log/erfc, v_rcp_f64 reciprocal) through
aq_special_eval_device, against the host build of the same header and against mpmath."""
import numpy as np
import pytest

from atlasqtl_amd import _lib

pytestmark = pytest.mark.gpu


def evd(which, x, x2=None):
    L = _lib.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    a2 = np.ascontiguousarray(x2, dtype=np.float64) if x2 is not None else None
    rc = L.aq_special_eval_device(which, _lib.as_dp(x), _lib.as_dp(a2) if a2 is not None else None, _lib.as_dp(out),
                                  x.size, 0)
    assert rc == 0, L.aq_last_error()
    return out


def test_device_and_host_builds_agree():
    from tests.test_special import ev
    x = np.concatenate([np.linspace(-40, 40, 2001), [0.0, 1e-300, -1e-300]])
    for which in (0, 4, 5, 6, 7, 8, 10, 11, 12, 13):
        h, d = ev(which, x), evd(which, x)
        err = np.max(np.abs(h - d) / np.maximum(np.abs(h), 1e-300))
        assert err < (2e-13 if which == 0 else 5e-15), (which, err)
    xp = np.linspace(0.01, 60, 1500)
    for which in (1, 9):
        h, d = ev(which, xp), evd(which, xp)
        assert np.max(np.abs(h - d) / np.abs(h)) < 5e-15, which
    xs = np.linspace(1e-6, 1.0, 500)
    assert np.max(np.abs(ev(2, xs) - evd(2, xs)) / np.abs(ev(2, xs))) < 5e-15
    a = np.linspace(0.05, 1.95, 500)
    assert np.max(np.abs(ev(3, xp[:500], a) - evd(3, xp[:500], a)) / np.abs(ev(3, xp[:500], a))) < 1e-13


def test_short_chain_sigmoid_on_device():
    import mpmath as mp
    mp.mp.dps = 60
    x = np.concatenate([np.linspace(-800, 800, 1601), np.linspace(-40, 40, 4001), [0.0, 1e-300, 1e4, -1e4, 744.9, 745.5, 1e300]])
    ref = np.array([float(1 / (1 + mp.exp(mp.mpf(float(v))))) for v in x])
    got = evd(13, x)
    assert np.max(np.abs(got - ref) / np.maximum(ref, 1e-300)) < 2e-15


def test_probit_tables_on_device():
    """Device build of the table-driven probit terms (what the helper wave of the sweep kernel evaluates, there from its LDS copy
    of the same table) against 60-digit mpmath, tolerances as in tests/test_special.py."""
    from tests.test_special import _probit_refs, probit_table_points, ev
    x = probit_table_points()
    A, imr1, imr0 = _probit_refs(x)
    gA, g1, g0 = evd(18, x), evd(19, x), evd(20, x)
    assert np.max(np.abs(gA - A) / np.maximum(np.abs(A), 1.0)) < 4e-15
    scale = np.maximum(np.maximum(np.abs(imr1), np.abs(imr0)), 1.0)
    assert np.max(np.abs(g1 - imr1) / scale) < 4e-15
    assert np.max(np.abs(g0 - imr0) / scale) < 4e-15
    assert np.all(g1 >= -x) and np.all(g0 <= -x)
    for w in (18, 19, 20):                                   # host and device builds of the same header
        h, d = ev(w, x), evd(w, x)
        assert np.max(np.abs(h - d) / np.maximum(np.abs(h), 1.0)) < 2e-15


def test_log_ndtr_pair_tables_on_device():
    """Device build of the ELBO pass's log Phi / log(1 - Phi) (tables + tail series) against 60-digit mpmath and against the host build."""
    from tests.test_special import _log_ndtr_pair_refs, probit_table_points, ev
    x = probit_table_points()
    lP, l1 = _log_ndtr_pair_refs(x)
    gP, g1 = evd(24, x), evd(25, x)
    assert np.max(np.abs(gP - lP) / np.maximum(np.abs(lP), 1.0)) < 4e-15
    assert np.max(np.abs(g1 - l1) / np.maximum(np.abs(l1), 1.0)) < 4e-15
    for w in (24, 25):
        h, d = ev(w, x), evd(w, x)
        assert np.max(np.abs(h - d) / np.maximum(np.abs(h), 1.0)) < 2e-15

