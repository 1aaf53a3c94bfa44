"""GPU: post-processing on the device against the oracle's restatement of R/summarise_output.R."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (130, 49), (2000, 300)])
def test_assign_bfdr_matches_oracle(shape):
    import atlasqtl_amd as A
    from oracle import atlasqtl_oracle as O
    rng = np.random.default_rng(5)
    m = rng.beta(0.05, 1.0, size=shape)
    m[rng.random(shape) < 0.1] = 1e-3            # many exact ties (order() keeps them in original order)
    if m.size > 5:
        m.flat[:3] = 1.0
    got, ref = A.assign_bFDR(m), O.assign_bFDR(m)
    assert got.shape == ref.shape
    # the device scan is a tree, R's cumsum is sequential: they differ by the rounding of the SEQUENTIAL sum (~n eps);
    # against an extended-precision cumsum the device result is good to 1e-13
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-15)
    v = m.reshape(-1, order="F")
    ind = np.argsort(-v, kind="stable")
    exact = np.empty(v.size, dtype=np.longdouble)
    exact[ind] = np.cumsum((1 - v[ind]).astype(np.longdouble)) / np.arange(1, v.size + 1)
    np.testing.assert_allclose(got.reshape(-1, order="F"), exact.astype(np.float64), rtol=1e-13, atol=1e-15)
    # ties straddling nothing: the ranks implied by the FDR values are the oracle's
    assert np.array_equal(np.argsort(got.reshape(-1, order="F"), kind="stable"),
                          np.argsort(ref.reshape(-1, order="F"), kind="stable"))


@pytest.mark.parametrize("fdr", [False, True])
def test_hotspot_sizes_operator_and_resident(fdr):
    import atlasqtl_amd as A
    from atlasqtl_amd.core import VbRun
    from oracle import atlasqtl_oracle as O
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    run = VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], (1, 2, 10), 0.1, 400, True, True).run()
    gam = run.result()["gam_vb"]
    thres = 0.05 if fdr else 0.5
    rs_ref, nb_ref = O.hotspot_sizes(gam, thres, fdr)
    rs_dev, nb_dev = run.hotspot_sizes(thres, fdr)         # from the device-resident gam_vb
    rs_op, nb_op = A.hotspot_sizes(gam, thres, fdr)        # operator on a host matrix
    run.close()
    assert nb_ref > 0
    np.testing.assert_array_equal(rs_dev, rs_ref)
    np.testing.assert_array_equal(rs_op, rs_ref)
    assert nb_dev == nb_ref == nb_op
