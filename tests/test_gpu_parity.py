"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances follow BASELINE.json's north star: mu_beta_vb within 1e-6
relative, ELBO trace within 1e-5 relative, same iteration count, same gam_vb ordering."""
import numpy as np
import pytest

from tests.util import make_problem, operator_inputs

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import atlasqtl_oracle as O
    return O


def _relerr(a, b, floor=1e-8):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


@pytest.mark.parametrize("p,q,c", [(40, 7, 1.0), (130, 33, 0.6), (16, 16, 1.0), (1, 1, 1.0)])
def test_core_dual_loop_matches_oracle(p, q, c):
    """aq_core_dual_loop vs the C restatement of src/coreLoop.cpp:38-86 (same Gram-space
    recursion, same order, FMA contraction off on both sides): agreement to rounding of exp/log."""
    import atlasqtl_amd as A
    O = _oracle()
    a = operator_inputs(p, q, seed=p + q, c=c)
    b = {k: (v.copy(order="F") if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in a.items()}
    rng = np.random.default_rng(1)
    si = rng.permutation(p).astype(np.int32)      # the reference accepts any visiting order
    sq = rng.permutation(q).astype(np.int32)
    O.core_dual_loop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                     a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"],
                     si, sq, c=c)
    A.coreDualLoop(b["cp_X"], b["cp_Y_X"], b["gam_vb"], b["log_Phi"], b["log_1mPhi"], b["log_sig2_inv_vb"],
                   b["log_tau_vb"], b["m1_beta"], b["cp_betaX_X"], b["mu_beta_vb"], b["sig2_beta_vb"], b["tau_vb"],
                   si, sq, c=c)
    for key in ("gam_vb", "mu_beta_vb", "m1_beta", "cp_betaX_X"):
        assert _relerr(b[key], a[key], floor=1e-6) < 1e-11, key


def test_core_dual_mis_loop_matches_oracle():
    import atlasqtl_amd as A
    O = _oracle()
    p, q = 37, 9
    a = operator_inputs(p, q, seed=5, mis=True, c=0.8)
    b = {k: (v.copy(order="F") if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in a.items()}
    O.core_dual_mis_loop(a["cp_X"], a["cp_X_rm"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"],
                         a["log_sig2_inv_vb"], a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"],
                         a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], a["sample_q"], c=0.8)
    A.coreDualMisLoop(b["cp_X"], b["cp_X_rm"], b["cp_Y_X"], b["gam_vb"], b["log_Phi"], b["log_1mPhi"],
                      b["log_sig2_inv_vb"], b["log_tau_vb"], b["m1_beta"], b["cp_betaX_X"], b["mu_beta_vb"],
                      b["sig2_beta_vb"], b["tau_vb"], b["shuffled_ind"], b["sample_q"], c=0.8)
    for key in ("gam_vb", "mu_beta_vb", "m1_beta", "cp_betaX_X"):
        assert _relerr(b[key], a[key], floor=1e-6) < 1e-11, key


def test_core_dual_loop_argument_errors():
    import atlasqtl_amd as A
    a = operator_inputs(8, 3)
    bad = a["shuffled_ind"].copy(); bad[0] = 99
    with pytest.raises(Exception, match="out of range"):
        A.coreDualLoop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                       a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"], a["sig2_beta_vb"],
                       a["tau_vb"], bad, a["sample_q"])
    # empty index vectors: the reference's loops do not execute -> nothing changes
    g0 = a["gam_vb"].copy()
    A.coreDualLoop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                   a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"], a["sig2_beta_vb"], a["tau_vb"],
                   np.zeros(0, dtype=np.int32), a["sample_q"])
    assert np.array_equal(g0, a["gam_vb"])


def _run_both(prob, anneal, maxit, tol=0.1, thinned=True):
    import atlasqtl_amd as A
    O = _oracle()
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], prob["q"], anneal, 1, tol, maxit, prob["list_hyper"],
                                        prob["list_init"], thinned_elbo_eval=thinned, debug=True, trace=tr,
                                        full_output=True)
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], prob["q"], anneal, 1, tol, maxit, 0,
                                        prob["list_hyper"], prob["list_init"], full_output=True,
                                        thinned_elbo_eval=thinned, debug=True)
    return ref, got, tr


def _check_state(ref, got, rtol_mu=1e-6):
    # mu_beta_vb within 1e-6 relative (north star); entries with |mu| < 1e-8 compared absolutely
    assert _relerr(got["mu_beta_vb"], ref["mu_beta_vb"], floor=1e-8) < rtol_mu
    assert np.max(np.abs(got["gam_vb"] - ref["gam_vb"])) < 1e-8
    assert _relerr(got["theta_vb"], ref["theta_vb"], floor=1e-6) < 1e-6
    assert _relerr(got["zeta_vb"], ref["zeta_vb"], floor=1e-6) < 1e-6
    assert _relerr(got["tau_vb"], ref["tau_vb"]) < 1e-8
    assert _relerr(got["lam2_inv_vb"], ref["lam2_inv_vb"], floor=1e-6) < 1e-6


@pytest.mark.parametrize("anneal", [None, (1, 2, 10), (2, 3, 5), (3, 2, 4)])
@pytest.mark.parametrize("nsweep", [1, 3])
def test_vb_first_sweeps_match_oracle(anneal, nsweep):
    """State after 1 and 3 sweeps (annealed and not): every S1-S20 step against the restated R driver."""
    prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)
    ref, got, _ = _run_both(prob, anneal, nsweep)
    assert got["it"] == ref["it"] == nsweep
    _check_state(ref, got)


@pytest.mark.parametrize("shape", [(100, 75, 20), (200, 500, 50), (90, 33, 17), (300, 130, 49)])
def test_vb_full_run_matches_oracle(shape):
    """Whole run to convergence: same iteration count, ELBO trace within 1e-5 relative
    (observed ~1e-12), final state within 1e-6, identical PPI ordering above the tie tolerance."""
    n, p, q = shape
    prob = make_problem(n, p, q, p_act=min(10, p // 3), prob_assoc=1.0 if q <= 20 else 0.2)
    ref, got, tr = _run_both(prob, (1, 2, 10), 1000)
    assert got["converged"] and ref["converged"]
    assert got["it"] == ref["it"]
    its, lbs = got["elbo_trace"]
    ref_tr = [(r["it"], r["lb"]) for r in tr if r["lb"] is not None]
    assert list(its) == [t[0] for t in ref_tr]
    assert np.max(np.abs(lbs - np.array([t[1] for t in ref_tr])) / np.abs(lbs)) < 1e-5
    assert np.all(np.diff(lbs) > -np.sqrt(np.finfo(float).eps))        # the reference's monotonicity check
    _check_state(ref, got)
    # gam_vb ordering: stable sort, entries closer than 1e-9 count as ties
    g_ref, g_got = ref["gam_vb"].ravel(), got["gam_vb"].ravel()
    order = np.argsort(-g_ref, kind="stable")
    top = order[: min(200, order.size)]
    assert np.all(np.diff(g_got[top]) <= 1e-9)


def test_vb_padding_edges():
    """p and q not multiples of 16, n not a multiple of 64, single trait."""
    for (n, p, q) in [(70, 17, 1), (65, 16, 16), (130, 47, 3)]:
        prob = make_problem(n, p, q, p_act=min(4, p // 2), prob_assoc=1.0)
        ref, got, _ = _run_both(prob, None, 2)
        _check_state(ref, got)


@pytest.mark.parametrize("tt,stagger", [(1, 0), (2, 0), (2, 4)])
@pytest.mark.parametrize("chain", [0, 5])
@pytest.mark.parametrize("shape", [(1000, 208, 40), (1008, 130, 17)])
def test_bench_kernel_instance_matches_oracle(shape, chain, tt, stagger, monkeypatch):
    """The instances bench.py measures (n = 1000 -> 63 residual tiles): with two trait tiles per workgroup the host's geometry --
    since the end of round 3 aq_core_sweep_la_kernel<9,9,*,2,false,9>: 9 tiles on every matrix wave and 9 on the recurrence wave,
    helper wave at raised priority; C3 runs this one (q = 40 pads to 4 tiles, the last one all padding; q = 17 to 2 tiles, the
    second one partly); <10,9,*,2> and <10,10,*,2> through test_recurrence_wave_tile_counts_match_oracle below -- and the one-tile
    instance (the trait shards of a multi-GPU run), SIMD partners in step and staggered, plain launch and 5 chained SNP
    segments (the bench's 13: tests/test_gpu_bigp.py), against the oracle over a whole annealed run."""
    monkeypatch.setenv("AQ_CHAIN", str(chain))
    monkeypatch.setenv("AQ_TT", str(tt))
    monkeypatch.setenv("AQ_STAGGER", str(stagger))
    n, p, q = shape
    prob = make_problem(n, p, q, p_act=8, prob_assoc=0.3)
    ref, got, tr = _run_both(prob, (1, 2, 10), 1000)
    assert got["core_kernel"] == 0
    assert got["it"] == ref["it"] and got["converged"] == ref["converged"]
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(got["elbo_trace"][1], lref, rtol=1e-9)
    _check_state(ref, got)


@pytest.mark.parametrize("tt,nt3", [(1, 3), (1, 6), (1, 9), (2, 9), (2, 3)])
@pytest.mark.parametrize("chain", [0, 3])
def test_recurrence_wave_tile_counts_match_oracle(tt, nt3, chain, monkeypatch):
    """The instances in which the recurrence wave owns 3 / 6 / 9 residual tiles (geometries NT/NT/3, NT/NT-1/6, NT/NT/9 of the
    same 63 tiles at n = 1000), for one and two trait tiles per workgroup -- with one tile the next block's cross-block
    correction is accumulated inside the chain, with two the helper wave forms it -- plain and chained: whole annealed run."""
    monkeypatch.setenv("AQ_TT", str(tt))
    monkeypatch.setenv("AQ_NT3", str(nt3))
    monkeypatch.setenv("AQ_CHAIN", str(chain))
    monkeypatch.setenv("AQ_LA_NOSPLIT", "1")
    prob = make_problem(1000, 150, 40, p_act=8, prob_assoc=0.3)
    ref, got, tr = _run_both(prob, (1, 2, 10), 1000)
    assert got["core_kernel"] == 0 and got["it"] == ref["it"] and got["converged"] == ref["converged"]
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(got["elbo_trace"][1], lref, rtol=1e-9)
    _check_state(ref, got)


@pytest.mark.parametrize("tt", [1, 2])
@pytest.mark.parametrize("n", [20, 100, 200, 330, 512, 600, 768, 860, 912, 1000, 1024, 1040])
def test_look_ahead_kernel_every_tile_count_matches_oracle(n, tt, monkeypatch):
    """Residual-tile geometries NT/NT2 = 1/1 ... 11/11 of the look-ahead kernel as the host picks them (two tiles per workgroup: 9 / 8 / 6 at
    n = 860 and 912, 9 / 9 / 9 at 1000, 10 / 10 / 9 at 1024 and 1040, all with the helper wave at raised priority; n = 1040: 65 of the 66 tiles one workgroup
    holds; beyond n = 1056 the sample axis is split over several workgroups, tests/test_gpu_sharded.py), one and two trait
    tiles per workgroup, 12 sweeps each incl. the ladder and two ELBO evaluations."""
    monkeypatch.setenv("AQ_TT", str(tt))
    prob = make_problem(n, 100, 24, p_act=6, prob_assoc=0.5)
    ref, got, tr = _run_both(prob, (1, 2, 10), 12, thinned=False)
    assert got["core_kernel"] == 0 and got["it"] == ref["it"] == 12
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(got["elbo_trace"][1], lref, rtol=1e-9)
    _check_state(ref, got)


def test_vb_elbo_not_monotone_is_an_error():
    """The reference stop()s when the ELBO decreases (debug <- TRUE, R/atlasqtl_global_local_core.R:359-360).  Drive the
    library into that branch: capture the loop state between two sweeps, raise the stored previous ELBO far above anything
    the next evaluation can reach, restore it -> the next evaluation must fail with AQ_ERR_NUMERIC and the handle must stay
    failed; the same blob with debug = FALSE runs on (the reference only checks under debug)."""
    import struct
    from atlasqtl_amd._lib import AtlasqtlHipError
    from atlasqtl_amd.core import VbRun
    prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)

    def mk(debug):
        return VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], None, 0.1, 60, False, debug)

    a = mk(True)
    a.run_sweeps(5)
    blob = a.get_state().copy()
    a.run()                                                   # a healthy run passes the check
    _, lbs = a.elbo_trace()
    assert np.all(np.diff(lbs) > -1.5e-8)
    a.close()
    # AqStateHeader: u64 magic, 18 x i32, then doubles c, c_s, sig2_zeta, lb_new, lb_old
    off_lb_new = 8 + 18 * 4 + 3 * 8
    assert struct.unpack_from("<d", blob, off_lb_new)[0] == lbs[4]
    struct.pack_into("<d", blob, off_lb_new, 1e12)
    b = mk(True).set_state(blob)
    with pytest.raises(AtlasqtlHipError, match=r"\[4\].*ELBO not increasing monotonically"):
        b.run()
    with pytest.raises(AtlasqtlHipError, match="failed state"):
        b.run()
    b.close()
    c = mk(False).set_state(blob)
    c.run()
    assert c.status()["it"] > 5
    c.close()
