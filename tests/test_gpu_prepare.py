"""GPU: the device-side input construction (aq_prepare_data: scale(X), constant / duplicate-column removal,
centring of Y with NA -- R/prepare_atlasqtl.R:57-83, R/utils.R:276-343) against its NumPy restatement
(oracle/prepare_oracle.py), with fp64 and with int8 dosage input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(n, p, q, seed, na=0.0):
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.05, 0.5, size=p)
    G = rng.binomial(2, maf[None, :], size=(n, p)).astype(np.int8)
    G[:, 3] = 1                                   # constant
    G[:, p - 2] = 2                               # constant
    G[:, 7] = G[:, 1]                             # duplicates (the later copy goes)
    G[:, p - 1] = G[:, 4]
    G[:, 11] = G[:, 1]                            # a third copy of column 1
    Y = rng.normal(size=(n, q))
    if na > 0:
        Y[rng.random(Y.shape) < na] = np.nan
    return G, Y


@pytest.mark.parametrize("n,p,q,na", [(50, 16, 3, 0.0), (333, 100, 7, 0.1), (1000, 257, 5, 0.3), (5000, 40, 2, 0.05)])
def test_device_prepare_matches_oracle(n, p, q, na):
    from atlasqtl_amd import prepare as P
    from oracle import prepare_oracle as PO
    G, Y = _case(n, p, q, seed=n + p, na=na)
    Xs_ref, Yc_ref, cst_ref, coll_ref = PO.prepare_xy(Y, G.astype(np.float64))
    rm_ref = cst_ref.copy()
    rm_ref[~cst_ref] = coll_ref
    outs = []
    for X in (G.astype(np.float64), G):                       # fp64 and int8 dosage input
        prep, cst, coll, dup = P.prepare_on_device(Y, X)
        np.testing.assert_array_equal(cst, cst_ref)
        np.testing.assert_array_equal(cst | coll, rm_ref)
        assert set(np.where(coll)[0]) == {7, 11, p - 1} and dup[7] == 1 and dup[11] == 1 and dup[p - 1] == 4
        Xs = prep.X_host()
        assert Xs.shape == Xs_ref.shape == (n, p - 5)
        np.testing.assert_allclose(Xs, Xs_ref, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose((Xs ** 2).sum(0), n - 1.0, rtol=1e-12)        # diag(X'X) = n - 1
        np.testing.assert_allclose(prep.Y, Yc_ref, rtol=1e-12, atol=1e-14, equal_nan=True)
        outs.append(Xs)
        prep.close()
    np.testing.assert_array_equal(outs[0], outs[1])           # the two input types give the same bits


@pytest.mark.parametrize("val", [0.1, 1.0 / 3.0, 2.7])
def test_device_prepare_constant_non_dyadic_column(val):
    """R centres a constant column to exactly 0 -> NaN -> removed, whatever the rounding of its mean (R/utils.R:276-302)."""
    from atlasqtl_amd import prepare as P
    rng = np.random.default_rng(3)
    X = rng.binomial(2, 0.3, size=(333, 6)).astype(float)
    X[:, 2] = val
    prep, cst, coll, _ = P.prepare_on_device(rng.normal(size=(333, 2)), X)
    assert list(np.where(cst)[0]) == [2] and not coll.any() and prep.p == 5
    prep.close()


def test_device_prepare_guards():
    from atlasqtl_amd import prepare as P
    rng = np.random.default_rng(0)
    X = rng.normal(size=(40, 5)); Y = rng.normal(size=(40, 3))
    Yn = Y.copy(); Yn[:, 1] = np.nan
    with pytest.raises(P.AtlasqtlError, match="97.5% missing"):
        P.prepare_on_device(Yn, X)
    Yn = np.full_like(Y, np.nan); Yn[0, :] = 1.0          # 1 of 40 observed everywhere: 2.5 % < 5 % overall (R/prepare_atlasqtl.R:39)
    with pytest.raises(P.AtlasqtlError, match="Too few non-NA"):
        P.prepare_on_device(Yn, X)
    with pytest.raises(P.AtlasqtlError, match="at least 1 non-constant"):
        P.prepare_on_device(Y, np.ones((40, 3)))


def test_prepared_data_feeds_the_run_without_a_host_copy_of_x():
    """atlasqtl() with int8 dosages: the standardised matrix is built and stays on the GPU; the run equals the one from the
    same matrix handed over as host fp64."""
    import atlasqtl_amd as A
    from atlasqtl_amd import synth
    d = synth.simulate(200, 130, 24, p_act=8, seed=5, maf=0.25, prob_assoc=0.4)
    G = d["X"].astype(np.int8)
    assert np.array_equal(G.astype(float), d["X"])
    a = A.atlasqtl(Y=d["Y"], X=G, p0=(3, 9), user_seed=4, verbose=0, full_output=True)
    b = A.atlasqtl(Y=d["Y"], X=d["X"].astype(float), p0=(3, 9), user_seed=4, verbose=0, full_output=True)
    assert a.converged and a.it == b.it and a.lb_opt == b.lb_opt
    np.testing.assert_array_equal(a.gam_vb, b.gam_vb)


@pytest.mark.parametrize("mask", ["0", "0x3"])
def test_duplicates_found_behind_hash_collisions(mask, monkeypatch):
    """rm_collinear_ = duplicated(mat, MARGIN = 2) (R/utils.R:304-343) keeps the first of every group of identical columns.  The
    device finds candidates by a 128-bit column hash and confirms them bitwise; with the hashes truncated (test hook) DIFFERENT
    columns share a hash class, and a duplicate of a column that is not the first of its class must still be found."""
    from atlasqtl_amd.prepare import prepare_on_device
    from oracle import prepare_oracle as PO
    monkeypatch.setenv("AQ_PREP_HASH_MASK", mask)
    rng = np.random.default_rng(5)
    n = 60
    base = rng.binomial(2, 0.3, size=(n, 7)).astype(np.float64)
    #            0        1        2        3 = dup of 1   4 = dup of 2   5        6 = dup of 0   7 = dup of 5   8 = dup of 1
    X = np.column_stack([base[:, 0], base[:, 1], base[:, 2], base[:, 1], base[:, 2], base[:, 3], base[:, 0], base[:, 3], base[:, 1]])
    Y = rng.normal(size=(n, 3))
    prep, cst, coll, dup_of = prepare_on_device(Y, X, 0)
    try:
        _, _, cst_o, coll_o = PO.prepare_xy(Y, X)
        assert not cst.any() and not cst_o.any()
        assert list(np.where(coll)[0]) == [3, 4, 6, 7, 8] == list(np.where(coll_o)[0])
        assert list(dup_of[[3, 4, 6, 7, 8]]) == [1, 2, 0, 5, 1]
    finally:
        prep.close()
