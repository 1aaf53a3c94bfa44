"""Host-side mirror of the reference's R pre-processing / hyper / init code (no GPU)."""
import numpy as np
import pytest

import atlasqtl_amd as A
from atlasqtl_amd import hyper_init as H
from atlasqtl_amd import prepare as P
from atlasqtl_amd import synth
from oracle import prepare_oracle as PO


def test_scale_and_removals():
    """The NumPy restatement of scale(X) / rm_constant_ / rm_collinear_ / centring (oracle/prepare_oracle.py: the checker of
    the device-side preparation, tests/test_gpu_prepare.py)."""
    rng = np.random.default_rng(0)
    X = rng.binomial(2, 0.3, size=(50, 8)).astype(float)
    X[:, 3] = 1.0                      # constant  -> NaN after scale -> removed (R/utils.R:278)
    X[:, 6] = X[:, 1]                  # duplicate -> removed, later one dropped (R/utils.R:307)
    Y = rng.normal(size=(50, 4))
    Xs, Yc, bool_cst, bool_coll = PO.prepare_xy(Y, X)
    assert Xs.shape == (50, 6)
    assert list(np.where(bool_cst)[0]) == [3] and list(np.where(bool_coll)[0]) == [5]      # column 6 of 8 = 5th non-constant
    np.testing.assert_allclose((Xs ** 2).sum(0), 49.0)          # diag(X'X) = n - 1 (note N1)
    np.testing.assert_allclose(Yc.mean(0), 0, atol=1e-14)


@pytest.mark.parametrize("val", [0.1, 0.3, 1.0 / 3.0, 2.7])
@pytest.mark.parametrize("n", [50, 333, 1000])
def test_constant_non_dyadic_column_is_removed(val, n):
    """R's scale() centres a constant column to exactly 0 (long-double colMeans) -> 0/0 = NaN -> rm_constant_ drops it
    (R/utils.R:276-302).  A pairwise mean that is one ulp off would leave a finite +-0.99 column behind."""
    rng = np.random.default_rng(n)
    X = rng.binomial(2, 0.3, size=(n, 5)).astype(float)
    X[:, 2] = val
    Xs = PO.scale_columns(X)
    assert np.isnan(Xs[:, 2]).all()
    Xk, _, bool_cst, bool_coll = PO.prepare_xy(rng.normal(size=(n, 3)), X)
    assert list(np.where(bool_cst)[0]) == [2] and not bool_coll.any()
    np.testing.assert_allclose((Xk ** 2).sum(0), n - 1.0)


def test_input_guards():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(40, 5)); Y = rng.normal(size=(40, 3))
    with pytest.raises(A.AtlasqtlError, match="same number of samples"):
        P.prepare_data_(Y[:30], X, 0.1, 10, None, 0, None, None)
    Xn = X.copy(); Xn[0, 0] = np.nan
    with pytest.raises(A.AtlasqtlError):
        P.prepare_data_(Y, Xn, 0.1, 10, None, 0, None, None)          # X must be NA-free (R/prepare_atlasqtl.R:19)
    Yn = Y.copy(); Yn[:, 1] = np.nan
    with pytest.raises(A.AtlasqtlError, match="97.5% missing"):       # (checked on the host before anything goes to the GPU)
        P.prepare_data_(Yn, X, 0.1, 10, None, 0, None, None)
    with pytest.raises(A.AtlasqtlError, match="positive"):
        P.prepare_data_(Y, X, 0.0, 10, None, 0, None, None)
    with pytest.raises(A.AtlasqtlError, match="natural"):
        P.prepare_data_(Y, X, 0.1, 2.5, None, 0, None, None)


@pytest.mark.parametrize("anneal,msg", [((4, 2, 10), "spacing scheme"), ((1, 1.2, 10), "temperature very small"),
                                        ((1, 2, 2000), "grid size very large"), ((1, 2), "anneal")])
def test_check_annealing(anneal, msg):
    with pytest.raises(A.AtlasqtlError, match=msg):
        P.check_annealing_(anneal)
    P.check_annealing_(None)
    P.check_annealing_((2, 1.5, 3))


def test_set_hyper_and_init_validation():
    h = A.set_hyper(3, 5, eta=1.0, kappa=[1, 2, 3], n0=-2.0, nu=0.01, rho=1.0, t02=0.1)
    assert h["eta"].shape == (3,) and h["m0"] == 0 and h["A2_inv"] == 1 and h.cls == "hyper"
    with pytest.raises(A.AtlasqtlError):
        A.set_hyper(3, 5, eta=-1.0, kappa=1.0, n0=0.0, nu=0.01, rho=1.0, t02=0.1)
    rng = np.random.default_rng(0)
    with pytest.raises(A.AtlasqtlError, match="between 0 and 1"):
        A.set_init(3, 5, rng.uniform(1, 2, (5, 3)), rng.normal(size=(5, 3)), 1.0, np.ones(3), np.ones(5), np.ones(3),
                   np.zeros(5), np.zeros(3))
    li = A.set_init(3, 5, rng.uniform(0, 1, (5, 3)), rng.normal(size=(5, 3)), 1.0, np.ones(3), np.ones(5), np.ones(3),
                    np.zeros(5), np.zeros(3))
    assert li.cls == "init" and li["gam_vb"].shape == (5, 3)


def test_auto_hyper_matches_prior_moments():
    """t02, n0 solve E[#active] = p0[1], Var = p0[2] (R/set_hyper_init.R:161-179, R/utils.R:218-242)."""
    Y = np.random.default_rng(0).normal(size=(80, 6))
    h = H.auto_set_hyper_(Y, 500, (5, 25))
    mu, t02 = h["n0"][0], h["t02"]
    assert abs(500 * H.E_Phi_X(mu, t02) - 5) < 1e-8
    assert abs(H.get_V_p_t(mu, t02, 500) - 25) < 1e-6
    assert h["nu"] == 1e-2 and h["rho"] == 1 and np.all(h["kappa"] == 1)
    with pytest.raises(A.AtlasqtlError, match="No hyperparameter values"):
        H.auto_set_hyper_(Y, 10, (5, 1e6))


def test_auto_init_shapes_and_seed():
    Y = np.random.default_rng(0).normal(size=(80, 6))
    a = H.auto_set_init_(Y, 30, (5, 25), 6, 7)
    b = H.auto_set_init_(Y, 30, (5, 25), 6, 7)
    assert a["gam_vb"].shape == (30, 6) and np.all((a["gam_vb"] >= 0) & (a["gam_vb"] <= 1))
    np.testing.assert_array_equal(a["mu_beta_vb"], b["mu_beta_vb"])
    assert a["sig2_theta_vb"].shape == (30,) and np.all(a["sig2_beta_vb"] > 0)


def test_list_dimension_checks():
    d = synth.simulate(60, 20, 4, p_act=3, seed=1)
    Xs, Yc, bool_cst, bool_coll = PO.prepare_xy(d["Y"], d["X"])
    bool_rmvd_x = bool_cst.copy()
    bool_rmvd_x[~bool_cst] = bool_coll
    p = Xs.shape[1]
    bad = A.set_hyper(5, len(bool_rmvd_x), 1.0, 1.0, -2.0, 0.01, 1.0, 0.1)
    with pytest.raises(A.AtlasqtlError, match=r"dimensions \(q\)"):
        H.prepare_list_hyper_(bad, Yc, p, (2, 4), bool_rmvd_x)
    with pytest.raises(A.AtlasqtlError, match="must be an object of class"):
        H.prepare_list_hyper_({"q_hyper": 4}, Yc, p, (2, 4), bool_rmvd_x)


def test_operand_prefetch_protocol_is_consistent():
    """The request / wait protocol of the look-ahead kernel's deeper operand prefetch (atlasqtl_amd/csrc/aq_core_sweep_la.h:
    aq_req_index / aq_req_issued, D buffers per stream, tile k in buffer k % D, every wait = vmcnt(2 x requests issued after the
    wanted one), loads complete in order), simulated on the host for every tile count and both depths over several phases:
    every tile an MFMA consumes has returned by then and still sits in its buffer."""
    def req_index(D, last, stream, k):
        n = 0
        for j in range(0, D - 1):
            if stream == 0 and k == j:
                return n
            n += 1
            if j <= D - 3:
                if stream == 1 and k == j:
                    return n
                n += 1
        for t in range(0, last + 1):
            if t + D - 1 <= last:
                if stream == 0 and k == t + D - 1:
                    return n
                n += 1
            if t + D - 2 <= last:
                if stream == 1 and k == t + D - 2:
                    return n
                n += 1
        return -1

    def req_issued(D, last, t):
        return (D - 1) + (D - 2) + sum((s + D - 1 <= last) + (s + D - 2 <= last) for s in range(t + 1))

    def simulate(NTC, D, phases=4):
        last, queue, buf = NTC - 1, [], {}

        def issue(stream, ph, k):
            queue.append((stream, ph, k))
            buf[(stream, k % D)] = [ph, k, False]

        def wait(vmcnt):
            keep = vmcnt // 2
            assert 0 <= vmcnt < 64
            if keep < len(queue):
                for (st, ph, k) in queue[:len(queue) - keep]:
                    b = buf[(st, k % D)]
                    if b[0] == ph and b[1] == k:
                        b[2] = True
                del queue[:len(queue) - keep]

        def use(stream, ph, k):
            assert buf[(stream, k % D)] == [ph, k, True], (NTC, D, stream, ph, k, buf[(stream, k % D)])

        def dangling(ph, skip_first):
            for j in range(0, D - 1):
                if not (skip_first and j == 0):
                    issue(0, ph, j)
                if j <= D - 3:
                    issue(1, ph, j)

        dangling(0, False)
        for ph in range(phases):
            for t in range(NTC):
                if t + D - 1 <= last:
                    issue(0, ph, t + D - 1)
                if t + D - 2 <= last:
                    issue(1, ph, t + D - 2)
                issued = req_issued(D, last, t)
                wait(2 * (issued - 1 - req_index(D, last, 0, t)))
                use(0, ph, t)
                if t >= 1:
                    wait(2 * (issued - 1 - req_index(D, last, 1, t - 1)))
                    use(1, ph, t - 1)
                if t == last:
                    issue(0, ph + 1, 0)
                    wait(2 * (issued - 1 - req_index(D, last, 1, last) + 1))
                    use(1, ph, last)
                    dangling(ph + 1, True)

    for D in (3, 4):
        for NTC in range(4, 19):
            simulate(NTC, D)
