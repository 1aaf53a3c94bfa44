"""GPU: the user-level atlasqtl() mirror end to end (argument surface of R/atlasqtl.R:179-184)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n=120, p=40, q=12, seed=3):
    from atlasqtl_amd import synth
    d = synth.simulate(n, p, q, p_act=6, seed=seed, maf=0.25, prob_assoc=0.6)
    return d["X"].astype(float), d["Y"], d


def test_atlasqtl_defaults_converge_like_reference_test():
    """The reference's only test (tests/testthat/test_convergence.R): atlasqtl(Y, X, p0) converges."""
    import atlasqtl_amd as A
    X, Y, d = _data(100, 75, 20, seed=123)
    vb = A.atlasqtl(Y=Y, X=X, p0=(5, 25), user_seed=1, verbose=0)
    assert vb.converged is True
    assert vb.gam_vb.shape == (vb.p, vb.q) and vb.beta_vb.shape == vb.gam_vb.shape
    assert np.all((vb.gam_vb >= 0) & (vb.gam_vb <= 1))
    top = set(np.argsort(-vb.gam_vb.sum(1))[:6])
    assert len(top & set(d["act_x"])) >= 4


def test_atlasqtl_matches_oracle_through_the_whole_wrapper():
    """Same pre-processing, hyper-parameters and initial values on both sides -> same run."""
    import atlasqtl_amd as A
    from atlasqtl_amd import hyper_init as H
    from atlasqtl_amd import prepare as P
    from oracle import atlasqtl_oracle as O
    X, Y, _ = _data()
    X[:, 5] = 2.0                 # constant column   -> removed (R/utils.R:276-302)
    X[:, 9] = X[:, 2]             # duplicated column -> removed (R/utils.R:304-343)
    vb = A.atlasqtl(Y=Y, X=X, p0=(3, 9), user_seed=7, verbose=0, save_hyper=True, save_init=True, full_output=True,
                    add_collinear_back=False)
    assert vb.rmvd_cst_x == ["Cov_x_6"] and vb.rmvd_coll_x == {"Cov_x_10": "Cov_x_3"}
    dat = P.prepare_data_(Y, X, 0.1, 1000, None, 0, None, None)
    ref = O.atlasqtl_global_local_core_(dat["Y"], dat["X"].X_host(), Y.shape[1], (1, 2, 10), 1, 0.1, 1000, vb.list_hyper,
                                        vb.list_init, full_output=True)
    assert vb.it == ref["it"] and vb.converged == ref["converged"]
    assert abs(vb.lb_opt - ref["lb_opt"]) <= 1e-9 * abs(ref["lb_opt"])
    np.testing.assert_allclose(vb.gam_vb, ref["gam_vb"], atol=1e-9)
    np.testing.assert_allclose(vb.beta_vb, ref["beta_vb"], rtol=1e-6, atol=1e-11)
    # save_init returns the TRUE initial values (the reference's in-place kernel clobbers them, R/atlasqtl.R:314)
    li2 = H.auto_set_init_(dat["Y"], dat["X"].shape[1], (3, 9), Y.shape[1], 7)
    np.testing.assert_array_equal(vb.list_init["gam_vb"], li2["gam_vb"])
    # collinear add-back duplicates the kept column's rows
    vb2 = A.atlasqtl(Y=Y, X=X, p0=(3, 9), user_seed=7, verbose=0, add_collinear_back=True)
    assert vb2.gam_vb.shape[0] == vb.gam_vb.shape[0] + 1
    np.testing.assert_array_equal(vb2.gam_vb[8], vb2.gam_vb[2])     # Cov_x_10 (index 8 after the constant is gone) = Cov_x_3


@pytest.mark.parametrize("anneal,thinned", [(None, True), ((3, 2, 5), False), ((2, 4, 8), True)])
def test_atlasqtl_options(anneal, thinned):
    import atlasqtl_amd as A
    from atlasqtl_amd import prepare as P
    from oracle import atlasqtl_oracle as O
    X, Y, _ = _data(90, 30, 9, seed=11)
    vb = A.atlasqtl(Y=Y, X=X, p0=(3, 9), anneal=anneal, user_seed=2, verbose=0, thinned_elbo_eval=thinned,
                    save_hyper=True, save_init=True)
    dat = P.prepare_data_(Y, X, 0.1, 1000, None, 0, None, None)
    tr = []
    ref = O.atlasqtl_global_local_core_(dat["Y"], dat["X"].X_host(), 9, anneal, 1, 0.1, 1000, vb.list_hyper, vb.list_init,
                                        thinned_elbo_eval=thinned, trace=tr)
    assert vb.it == ref["it"] and vb.converged
    assert abs(vb.diff_lb - ref["diff_lb"]) < 1e-6


def test_maxit_reached_warns_and_reports():
    import atlasqtl_amd as A
    X, Y, _ = _data()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        vb = A.atlasqtl(Y=Y, X=X, p0=(3, 9), maxit=12, user_seed=2, verbose=1)
    assert vb.converged is False and vb.it == 12
    assert any("Maximal number of iterations" in str(x.message) for x in w)     # R/atlasqtl_global_local_core.R:397


def test_argument_errors_keep_reference_wording():
    import atlasqtl_amd as A
    X, Y, _ = _data()
    with pytest.raises(A.AtlasqtlError, match="verbose argument must be set to 0, 1 or 2"):
        A.atlasqtl(Y, X, (3, 9), verbose=5)
    with pytest.raises(A.AtlasqtlError, match="spacing scheme"):
        A.atlasqtl(Y, X, (3, 9), anneal=(7, 2, 10), verbose=0)
    with pytest.raises(A.AtlasqtlError, match="same number of samples"):
        A.atlasqtl(Y[:50], X, (3, 9), verbose=0)


def _vbrun(prob, li=None, **kw):
    from atlasqtl_amd.core import VbRun
    return VbRun(prob["Y"], prob["X"], prob["list_hyper"], li if li is not None else prob["list_init"], (1, 2, 10), 0.1, 400,
                 True, True, **kw)


@pytest.mark.parametrize("na", [0.0, 0.06])
def test_state_roundtrip_continues_bit_identically(na):
    """aq_vb_get_state / aq_vb_set_state (the reference's checkpoint_ is write-only, R/utils.R:571-611):
    a second handle, created with DIFFERENT initial values, continues from the captured state to the very same bits."""
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3, na_frac=na)
    a = _vbrun(prob).run()
    ref_tr, ref = a.elbo_trace(), a.result(full_output=True)
    ref_st = a.status()
    a.close()
    for stop in (0, 7, 23):           # before the first sweep, inside the annealing ladder, on the thinned ELBO schedule
        b = _vbrun(prob)
        b.run_sweeps(stop)
        blob = b.get_state()
        b.close()
        li2 = dict(prob["list_init"])
        li2["gam_vb"] = np.asfortranarray(np.full_like(prob["list_init"]["gam_vb"], 0.3))
        li2["theta_vb"] = prob["list_init"]["theta_vb"] * 0.0
        c = _vbrun(prob, li2).set_state(blob)
        assert c.status()["it"] == stop
        c.run()
        st, tr, res = c.status(), c.elbo_trace(), c.result(full_output=True)
        c.close()
        assert st["it"] == ref_st["it"] and st["converged"] == ref_st["converged"] and st["lb_opt"] == ref_st["lb_opt"]
        np.testing.assert_array_equal(tr[0], ref_tr[0])
        np.testing.assert_array_equal(tr[1], ref_tr[1])
        for k in ("gam_vb", "mu_beta_vb", "theta_vb", "zeta_vb", "tau_vb", "lam2_inv_vb"):
            np.testing.assert_array_equal(res[k], ref[k])


def test_state_of_another_problem_is_refused():
    from atlasqtl_amd._lib import AtlasqtlHipError as AqError
    from tests.util import make_problem
    a = _vbrun(make_problem(200, 130, 49, p_act=10))
    blob = a.get_state()
    a.close()
    b = _vbrun(make_problem(200, 131, 49, p_act=10))
    with pytest.raises(AqError, match="different problem shape"):
        b.set_state(blob)
    with pytest.raises(AqError, match="not an atlasqtl-hip state blob"):
        b.set_state(np.zeros(4096, dtype=np.uint8))
    b.close()
    # same shape, but another trait shard of a sharded run / another scheme: refused as well (header v02)
    prob = make_problem(200, 130, 49, p_act=10)
    c = _vbrun(prob, q_total=98, trait_offset=49)
    with pytest.raises(AqError, match="different problem shape"):       # q_total differs
        c.set_state(blob)
    blob_c = c.get_state()
    c.close()
    d = _vbrun(prob, q_total=98, trait_offset=0)
    with pytest.raises(AqError, match="another trait shard"):
        d.set_state(blob_c)
    d.close()
    e = _vbrun(prob, scheme="global")
    with pytest.raises(AqError, match="another scheme"):
        e.set_state(blob)
    e.close()


def test_checkpoint_path_and_resume(tmp_path):
    """checkpoint_path as in R/atlasqtl.R:183: temporary outputs every `rate` iterations, the last two kept, all removed
    at the end (R/utils.R:571-627) -- plus the state files that make `resume_from` possible."""
    import os
    import atlasqtl_amd as A
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    args = (prob["Y"], prob["X"], 49, (1, 2, 10), 1, 0.1, 400, 0, prob["list_hyper"], prob["list_init"])
    ref = A.atlasqtl_global_local_core_(*args, full_output=True)
    got = A.atlasqtl_global_local_core_(*args, full_output=True, checkpoint_path=str(tmp_path), checkpoint_rate=10)
    assert got["it"] == ref["it"] > 30
    np.testing.assert_array_equal(got["gam_vb"], ref["gam_vb"])
    assert os.listdir(tmp_path) == []                                         # converged: checkpoint_clean_up_, states too
    # stopped at maxit before convergence: the last two state files stay and a later call resumes from them
    short = list(args)
    short[6] = 25                                                             # maxit
    part = A.atlasqtl_global_local_core_(*short, full_output=True, checkpoint_path=str(tmp_path), checkpoint_rate=10)
    assert part["it"] == 25 and not part["converged"]
    files = sorted(os.listdir(tmp_path))
    assert not [f for f in files if f.startswith("tmp_output_it_")]
    states = [f for f in files if f.startswith("hip_state_it_")]
    assert states == ["hip_state_it_10.npy", "hip_state_it_20.npy"]           # the last two
    again = A.atlasqtl_global_local_core_(*args, full_output=True, resume_from=os.path.join(tmp_path, states[1]))
    assert again["it"] == ref["it"]
    np.testing.assert_array_equal(again["gam_vb"], ref["gam_vb"])
    np.testing.assert_array_equal(again["elbo_trace"][1], ref["elbo_trace"][1])


def _device_init_list(prob, seed=20240607):
    li = dict(prob["list_init"])
    li["gam_vb"] = li["mu_beta_vb"] = None
    li["device_seed"], li["device_gam_mean"], li["device_gam_sd"] = seed, float(np.asarray(prob["list_hyper"]["n0"])[0]), 0.37
    return li


def test_device_generated_init_equals_the_oracle_stream_and_runs_like_it():
    """gam_vb / mu_beta_vb of auto_set_init_ (R/set_hyper_init.R:385-387) drawn on the device from the
    Philox stream; the oracle restates the stream, so 'same (X, Y), same init' holds on both sides."""
    import atlasqtl_amd as A
    from atlasqtl_amd.core import VbRun
    from oracle import atlasqtl_oracle as O
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    li = _device_init_list(prob)
    run = VbRun(prob["Y"], prob["X"], prob["list_hyper"], li, (1, 2, 10), 0.1, 400, True, True)
    run.run_sweeps(0)
    res0 = run.result(full_output=True)
    run.close()
    g_ref, m_ref = O.philox_init(li["device_seed"], 130, 49, li["device_gam_mean"], li["device_gam_sd"])
    np.testing.assert_allclose(res0["gam_vb"], g_ref, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(res0["mu_beta_vb"], m_ref, rtol=1e-12, atol=1e-15)
    # the whole run from the device-drawn init == the oracle's run from the restated init
    li_ref = dict(prob["list_init"], gam_vb=g_ref, mu_beta_vb=m_ref)
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 49, (1, 2, 10), 1, 0.1, 400, prob["list_hyper"], li_ref, trace=tr,
                                        full_output=True)
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], 49, (1, 2, 10), 1, 0.1, 400, 0, prob["list_hyper"], li,
                                        full_output=True, debug=True)
    assert got["it"] == ref["it"]
    np.testing.assert_allclose(got["mu_beta_vb"], ref["mu_beta_vb"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], atol=1e-9)


def test_device_generated_init_does_not_depend_on_sharding():
    from atlasqtl_amd.core import VbRun
    from tests.util import make_problem
    prob = make_problem(120, 60, 40, p_act=6)
    li = _device_init_list(prob)
    full = VbRun(prob["Y"], prob["X"], prob["list_hyper"], li, None, 0.1, 10, True, True)
    full.run_sweeps(0)
    g_full = full.result()["gam_vb"]
    full.close()
    k0 = 16
    lh, li2 = dict(prob["list_hyper"]), dict(li)
    for k in ("eta", "kappa", "n0"):
        lh[k] = np.asarray(lh[k])[k0:]
    for k in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li2[k] = np.asarray(li2[k])[k0:]
    shard = VbRun(prob["Y"][:, k0:], prob["X"], lh, li2, None, 0.1, 10, True, True, q_total=40, trait_offset=k0)
    shard.run_sweeps(0)
    g_sh = shard.result()["gam_vb"]
    shard.close()
    np.testing.assert_array_equal(g_sh, g_full[:, k0:])


def _sharded_device_init_worker(rank, world, port, outdir):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import atlasqtl_amd as A
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    k0, k1 = (0, 32) if rank == 0 else (32, 49)
    lh, li = dict(prob["list_hyper"]), _device_init_list(prob)
    for k in ("eta", "kappa", "n0"):
        lh[k] = np.asarray(lh[k])[k0:k1]
    for k in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li[k] = np.asarray(li[k])[k0:k1]
    args = (prob["Y"][:, k0:k1], prob["X"], 49, (1, 2, 10), 1, 0.1, 1000, 0, lh, li)
    if rank == 0:   # without the offset the shards would silently all start from the draws of traits 0..q_local-1
        with pytest.raises(ValueError, match="trait_offset is required"):
            A.atlasqtl_global_local_core_(*args, process_group=dist.group.WORLD)
    out = A.atlasqtl_global_local_core_(*args, full_output=True, debug=True, process_group=dist.group.WORLD, trait_offset=k0)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), it=out["it"], lb=out["elbo_trace"][1], gam=out["gam_vb"], mu=out["mu_beta_vb"])
    dist.destroy_process_group()


def test_sharded_core_function_with_device_init_reproduces_single_gpu(tmp_path):
    """atlasqtl_global_local_core_(process_group=..., trait_offset=...) with device-drawn initial values: two trait shards
    (two processes on the one GPU, gloo-staged payloads) reproduce the single-process run -- the Philox counters are
    (SNP, GLOBAL trait)."""
    import socket
    import torch.multiprocessing as mp
    import atlasqtl_amd as A
    from tests.util import make_problem
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_sharded_device_init_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    one = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], 49, (1, 2, 10), 1, 0.1, 1000, 0, prob["list_hyper"],
                                        _device_init_list(prob), full_output=True, debug=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["it"]) == int(r1["it"]) == one["it"]
    np.testing.assert_allclose(r0["lb"], one["elbo_trace"][1], rtol=1e-10)
    np.testing.assert_allclose(np.concatenate([r0["mu"], r1["mu"]], axis=1), one["mu_beta_vb"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(np.concatenate([r0["gam"], r1["gam"]], axis=1), one["gam_vb"], atol=1e-10)


@pytest.mark.parametrize("env", [{"AQ_MIS_C": "3", "AQ_KERNEL": "3"}, {"AQ_CHAIN": "4"}, {"AQ_LA_C": "2"},
                                 {"AQ_LA_C": "2", "NA": "1"}, {"AQ_LA_C": "2", "AQ_LA_XHELPER": "1"}])
def test_expired_in_kernel_wait_is_reported_everywhere(env, monkeypatch):
    """A bounded wait that expires inside a sweep kernel (sample split: a partner's partial S; chained segments: the
    previous segment's residual) raises a device flag: results are invalid.  The flag is forced through the test hook;
    run, status, state and result getters must all report AQ_ERR_DEVICE afterwards."""
    import ctypes as C
    from atlasqtl_amd._lib import AtlasqtlHipError, lib
    from tests.util import make_problem
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    na = 0.04 if ("AQ_MIS_C" in env or "NA" in env) else 0.0
    prob = make_problem(300, 130, 49, p_act=8, prob_assoc=0.3, na_frac=na)
    run = _vbrun(prob)
    run.run_sweeps(3)
    assert run.status()["core_kernel"] == (3 if "AQ_KERNEL" in env else 0)
    assert lib().aq_vb_debug_raise_errflag(run.h) == 0
    for call in (run.status, run.get_state, run.result, lambda: run.run_sweeps(1)):
        with pytest.raises(AtlasqtlHipError, match=r"\[2\].*bounded wait"):
            call()
    run.close()


def test_atlasqtl_with_device_init_converges():
    import atlasqtl_amd as A
    X, Y, d = _data(100, 75, 20, seed=123)
    vb = A.atlasqtl(Y=Y, X=X, p0=(5, 25), user_seed=1, verbose=0, device_init=True)
    assert vb.converged is True
    top = set(np.argsort(-vb.gam_vb.sum(1))[:6])
    assert len(top & set(d["act_x"])) >= 4


def test_launch_plan_follows_the_problem_shape(monkeypatch):
    """aq_vb_status reports the launch plan of the core kernel.  Few trait groups and a long sample axis: the idle CUs share the
    samples (two workgroups per group); a short sample axis (the chain is the bound, a split only adds the exchange), Y with
    missing values, or AQ_LA_NOSPLIT=1: one workgroup per group; n beyond one workgroup's registers: always split."""
    from tests.util import make_problem

    def plan(n, p, q, **kw):
        run = _vbrun(make_problem(n, p, q, p_act=4, prob_assoc=0.5, **kw))
        st = run.status()
        run.close()
        return st["core_kernel"], st["split_parts"], st["tiles_per_group"], st["chain_segments"]

    assert plan(1000, 64, 24) == (0, 2, 1, 0)
    assert plan(200, 64, 24) == (0, 1, 1, 0)
    assert plan(1000, 64, 24, na_frac=0.05) == (0, 1, 1, 0)
    k, parts, tiles, chain = plan(2500, 40, 20)
    assert (k, tiles, chain) == (0, 1, 0) and parts >= 2
    monkeypatch.setenv("AQ_LA_NOSPLIT", "1")
    assert plan(1000, 64, 24) == (0, 1, 1, 0)


def test_env_overrides_are_reported(monkeypatch):
    """The AQ_* launch-plan hooks a handle was created under are visible through the C ABI (aq_vb_get_overrides): a host that
    inherits its environment can tell that the plan is not the library's own."""
    from atlasqtl_amd.core import VbRun
    from tests.util import make_problem
    prob = make_problem(100, 75, 20, p_act=5)
    r = VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], None, 0.1, 3)
    assert r.status()["overrides"] == ""
    r.close()
    monkeypatch.setenv("AQ_CHAIN", "3")
    monkeypatch.setenv("AQ_TT", "2")
    r = VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], None, 0.1, 3)
    ov = r.status()["overrides"].split()
    assert "AQ_CHAIN=3" in ov and "AQ_TT=2" in ov and len(ov) == 2
    r.close()
