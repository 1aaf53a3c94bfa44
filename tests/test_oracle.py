"""CPU tests of the oracle (test infrastructure): golden fixtures, the two independent
restatements of the inner update, Gram-space vs n-space, the reference's own test assertion
(convergence on its toy generator, tests/testthat/test_convergence.R:5-7) and its run-time
self check (ELBO monotone, R/atlasqtl_global_local_core.R:359-360)."""
import os

import numpy as np
import pytest

from oracle import atlasqtl_oracle as O
from oracle import sharded_oracle as S
from tests.util import make_problem, operator_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_vb(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    lh = {k[6:]: z[k] for k in z.files if k.startswith("hyper_")}
    li = {k[5:]: z[k] for k in z.files if k.startswith("init_")}
    for d in (lh, li):
        for k, v in list(d.items()):
            if v.ndim == 0:
                d[k] = float(v)
    anneal = tuple(z["anneal"]) if bool(z["has_anneal"]) else None
    return z, lh, li, anneal


@pytest.mark.parametrize("name", ["vb_toy_anneal", "vb_toy_noanneal", "vb_toy_missing", "vb_harmonic"])
def test_oracle_reproduces_golden(name):
    z, lh, li, anneal = load_vb(name)
    q = z["Y"].shape[1]
    tr = []
    out = O.atlasqtl_global_local_core_(z["Y"], z["X"], q, anneal, 1, 0.1, 1000, lh, li, trace=tr, full_output=True)
    assert out["it"] == int(z["out_it"]) and out["converged"] == bool(z["out_converged"])
    lbs = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(lbs, z["out_elbo_lb"], rtol=1e-11)
    np.testing.assert_allclose(out["gam_vb"], z["out_gam_vb"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out["mu_beta_vb"], z["out_mu_beta_vb"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(out["theta_vb"], z["out_theta_vb"], rtol=1e-8, atol=1e-12)
    assert np.all(np.diff(lbs) > -np.sqrt(np.finfo(float).eps))


@pytest.mark.parametrize("name,mis", [("op_core_dual_loop", False), ("op_core_dual_mis_loop", True)])
def test_operator_golden(name, mis):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    a = {k[3:]: np.array(z[k], order="F") if z[k].ndim == 2 else z[k] for k in z.files if k.startswith("in_")}
    c = float(a["c"])
    if mis:
        rms = [np.asfortranarray(m) for m in a["cp_X_rm"]]
        O.core_dual_mis_loop(a["cp_X"], rms, a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"],
                             float(a["log_sig2_inv_vb"]), a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"],
                             a["mu_beta_vb"], np.asfortranarray(a["sig2_beta_vb"]), a["tau_vb"], a["shuffled_ind"],
                             a["sample_q"], c=c)
    else:
        O.core_dual_loop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"],
                         float(a["log_sig2_inv_vb"]), a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"],
                         a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], a["sample_q"], c=c)
    for key in ("gam_vb", "mu_beta_vb", "m1_beta", "cp_betaX_X"):
        np.testing.assert_allclose(a[key], z["out_" + key], rtol=1e-12, atol=1e-14)


def test_two_restatements_of_inner_update_agree():
    """C restatement of src/coreLoop.cpp vs Python restatement of the reference's pure-R branch."""
    prob = make_problem(60, 30, 7, p_act=5, prob_assoc=1.0)
    a = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 7, (1, 2, 10), 1, 0.1, 30, prob["list_hyper"],
                                      prob["list_init"], inner="c", full_output=True)
    b = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 7, (1, 2, 10), 1, 0.1, 30, prob["list_hyper"],
                                      prob["list_init"], inner="pure", full_output=True)
    assert a["it"] == b["it"]
    np.testing.assert_allclose(a["gam_vb"], b["gam_vb"], atol=1e-12)
    np.testing.assert_allclose(a["mu_beta_vb"], b["mu_beta_vb"], rtol=1e-9, atol=1e-13)


def test_gram_space_equals_n_space():
    """cp_Y_X - (cp_betaX_X - m1 cp_X(j,j)) == x_j'(y - X beta) + (x_j'x_j) m1  to <= 1e-9."""
    a = operator_inputs(120, 11, n=80, seed=3)
    g = {k: (v.copy(order="F") if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in a.items()}
    O.core_dual_loop(g["cp_X"], g["cp_Y_X"], g["gam_vb"], g["log_Phi"], g["log_1mPhi"], g["log_sig2_inv_vb"],
                     g["log_tau_vb"], g["m1_beta"], g["cp_betaX_X"], g["mu_beta_vb"], g["sig2_beta_vb"], g["tau_vb"],
                     g["shuffled_ind"], g["sample_q"], c=1.0)
    X = np.asfortranarray(a["X"])
    R = np.asfortranarray(a["Y"] - X @ a["m1_beta"])
    O.nspace_loop(X, R, None, (X ** 2).sum(0), a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                  a["log_tau_vb"], a["m1_beta"], a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"], 1.0)
    np.testing.assert_allclose(a["mu_beta_vb"], g["mu_beta_vb"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a["gam_vb"], g["gam_vb"], atol=1e-12)
    np.testing.assert_allclose(R, a["Y"] - X @ a["m1_beta"], atol=1e-10)


def test_reference_toy_converges_and_finds_hotspots():
    """The reference's only test: vb$converged on n=100, p=75, q=20, p_act=10, p0=c(5,25)."""
    prob = make_problem(100, 75, 20, p_act=10, maf=0.2, prob_assoc=1.0)
    out = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 20, (1, 2, 10), 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"])
    assert out["converged"]
    top = set(np.argsort(-out["gam_vb"].sum(1))[:10])
    assert len(top & set(prob["truth"]["act_x"])) >= 8


def test_elbo_monotone_guard_fires(monkeypatch):
    """debug <- TRUE: a decreasing ELBO is an error (R/atlasqtl_global_local_core.R:359-360).  The driver restatement is
    run with an ELBO evaluator that drops by 1e9 on its third call: the guard must stop the loop right there, and with
    debug = FALSE (the reference's default outside debugging) the same run must carry on."""
    prob = make_problem(100, 40, 8, p_act=4, prob_assoc=1.0)
    real = O.elbo_global_local_
    calls = []

    def falling(*a, **k):
        calls.append(1)
        v = real(*a, **k)
        return v - 1e9 if len(calls) == 3 else v

    monkeypatch.setattr(O, "elbo_global_local_", falling)
    args = (prob["Y"], prob["X"], 8, None, 1, 0.1, 30, prob["list_hyper"], prob["list_init"])
    with pytest.raises(O.ElboNotMonotone, match="ELBO not increasing monotonically"):
        O.atlasqtl_global_local_core_(*args, thinned_elbo_eval=False, debug=True)
    assert len(calls) == 3
    calls.clear()
    out = O.atlasqtl_global_local_core_(*args, thinned_elbo_eval=False, debug=False)
    assert out["it"] > 3


def test_annealing_ladders():
    np.testing.assert_allclose(O.get_annealing_ladder_((1, 2, 10))[[0, -1]], [0.5, 1.0])
    lad = O.get_annealing_ladder_((2, 4, 5))
    np.testing.assert_allclose(1 / lad, np.linspace(4, 1, 5))
    np.testing.assert_allclose(O.get_annealing_ladder_((3, 2, 3)), [0.5, 0.75, 1.0])


def test_sharded_model_matches_driver_single_rank():
    """The decomposition the HIP path uses (n-space, Z = a + gam b, factorised ELBO) against the
    line-by-line driver: same iteration count, ELBO trace to 1e-10, state to 1e-8."""
    prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)
    for anneal in (None, (1, 2, 10)):
        tr_a, tr_b = [], []
        a = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 20, anneal, 1, 0.1, 1000, prob["list_hyper"],
                                          prob["list_init"], trace=tr_a, full_output=True)
        b = S.run_sharded(prob["Y"], prob["X"], 20, anneal, 0.1, 1000, prob["list_hyper"], prob["list_init"],
                          trace=tr_b)
        assert a["it"] == b["it"] and b["converged"]
        la = np.array([r["lb"] for r in tr_a if r["lb"] is not None])
        lb = np.array([r["lb"] for r in tr_b if r["lb"] is not None])
        np.testing.assert_allclose(lb, la, rtol=1e-10)
        np.testing.assert_allclose(b["mu_beta_vb"], a["mu_beta_vb"], rtol=1e-7, atol=1e-11)
        np.testing.assert_allclose(b["theta_vb"], a["theta_vb"], rtol=1e-8, atol=1e-11)


def test_assign_bfdr_restatement_properties():
    """R/summarise_output.R:207-223: the FDR of the top-ranked pair is 1 - its PPI, the values are non-decreasing along the
    ranking, and the last one is the mean of 1 - PPI."""
    from oracle import atlasqtl_oracle as O
    rng = np.random.default_rng(1)
    m = rng.beta(0.1, 1.0, size=(40, 9))
    f = O.assign_bFDR(m)
    v, fv = m.reshape(-1, order="F"), f.reshape(-1, order="F")
    order = np.argsort(-v, kind="stable")
    assert np.isclose(fv[order[0]], 1 - v[order[0]])
    assert np.all(np.diff(fv[order]) >= -1e-15)
    assert np.isclose(fv[order[-1]], np.mean(1 - v))
    rs, nb = O.hotspot_sizes(m, 0.5)
    assert nb == int((m > 0.5).sum()) and rs.shape == (40,)


def test_philox4x32_10_known_answers():
    """Known-answer vectors of the Random123 distribution (kat_vectors, philox4x32 with 10 rounds)."""
    from oracle import atlasqtl_oracle as O
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], key)
        assert tuple(int(g[0]) for g in got) == want
    from scipy import special as sp
    gam, mu = O.philox_init(12345, 400, 30, -2.0, 0.3)
    assert gam.shape == (400, 30) and 0 < gam.min() and gam.max() < 1
    assert abs(mu.mean()) < 0.05 and abs(mu.std() - 1) < 0.05                 # N(0, 1)
    z = (sp.ndtri(gam) + 2.0) / 0.3
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05                   # pnorm(N(-2, sd 0.3))
    g2, m2 = O.philox_init(12345, 400, 10, -2.0, 0.3, trait_offset=20)        # a shard reproduces its own columns
    np.testing.assert_array_equal(g2, gam[:, 20:])
    np.testing.assert_array_equal(m2, mu[:, 20:])


def test_compute_integral_hs_against_quadrature():
    """compute_integral_hs_ (R/utils.R:425-568), the closed forms behind df >= 5 of the horseshoe, restated in the
    oracle term by term and compared with numerical quadrature of int x^n (1 + a x)^-m exp(-b x) dx -- evidence about the
    REFERENCE's formulas, independent of any restatement of ours:
      * df = 5 (m = n = 3; m = 3, n = 2) and the m = 4, n = 3 list of df = 7 are right; they are differences of large terms, so
        digits go as L = b / a grows (the reference's own remark at :510);
      * the n = m = 4 list (:460-474, df = 7's numerator) does NOT add up to the integral, not even for small L.
    The device path reproduces the lists as written (parity with the reference is the contract), bugs included."""
    from scipy import integrate, special
    from oracle import atlasqtl_oracle as O
    err = {}
    for df in (3, 5, 7):
        e = (df + 1) // 2
        for L in (1e-3, 1e-2, 0.1, 0.5, 1.0, 3.0, 100.0):
            alpha, beta = float(df), L * df
            Q = float(np.exp(L) * special.exp1(L))
            for (m, n) in ((e, e), (e, e - 1)):
                ref, _ = integrate.quad(lambda x: x ** n * (1 + alpha * x) ** (-m) * np.exp(-beta * x), 0, np.inf,
                                        epsabs=0, epsrel=1e-12, limit=400)
                err[(df, m, n, L)] = abs(O.compute_integral_hs_(alpha, beta, m, n, Q) - ref) / abs(ref)
    for (df, m, n, L), v in err.items():
        if (m, n) == (4, 4):
            assert v > 1e-4, (df, m, n, L, v)                  # the reference's n = 4 list is not the integral
        elif L <= 3.0:
            assert v < 1e-10, (df, m, n, L, v)                 # right where the cancellation is mild
    assert err[(5, 3, 3, 100.0)] > 1e-9 and err[(7, 4, 3, 100.0)] > 1e-9      # digits lost at large L
