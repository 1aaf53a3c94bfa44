"""GPU: the real q-sharded HIP path on ONE box -- two processes share cuda:0, each holds half of the
traits, the all-reduce payloads travel through gloo (CPU-staged; RCCL needs one GPU per rank) --
against the single-process HIP run and the golden fixture."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import atlasqtl_amd as A
    from tests.util import make_problem
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    q = 49
    k0, k1 = (0, 32) if rank == 0 else (32, 49)
    lh, li = dict(prob["list_hyper"]), dict(prob["list_init"])
    for k in ("eta", "kappa", "n0"):
        lh[k] = np.asarray(lh[k])[k0:k1]
    for k in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li[k] = np.asarray(li[k])[k0:k1]
    for k in ("gam_vb", "mu_beta_vb"):
        li[k] = np.asarray(li[k])[:, k0:k1]
    out = A.atlasqtl_global_local_core_(prob["Y"][:, k0:k1], prob["X"], q, (1, 2, 10), 1, 0.1, 1000, 0, lh, li,
                                        full_output=True, debug=True, process_group=dist.group.WORLD)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), it=out["it"], lb=out["elbo_trace"][1], gam=out["gam_vb"],
             mu=out["mu_beta_vb"], theta=out["theta_vb"], zeta=out["zeta_vb"])
    dist.destroy_process_group()


@pytest.mark.parametrize("env", [{}, {"AQ_CHAIN": "3"}, {"AQ_LA_C": "2"}, {"AQ_LA_C": "3", "AQ_LA_XHELPER": "1"}])
def test_two_ranks_one_gpu_match_single_process(env, tmp_path, monkeypatch):
    """Two processes share the GPU (their kernels compete for CUs), plain and with the launches whose workgroups wait for one
    another -- chained SNP segments, sample split with either exchange wave: the hand-offs assume that a workgroup's
    predecessor / partners are resident or dispatched next; a violation would surface as
    AQ_ERR_DEVICE from the bounded waits, never as a hang."""
    import torch.multiprocessing as mp
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    import atlasqtl_amd as A
    from tests.util import make_problem
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    prob = make_problem(200, 130, 49, p_act=10, prob_assoc=0.3)
    one = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], 49, (1, 2, 10), 1, 0.1, 1000, 0, prob["list_hyper"],
                                        prob["list_init"], full_output=True, debug=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["it"]) == int(r1["it"]) == one["it"]
    np.testing.assert_allclose(r0["lb"], one["elbo_trace"][1], rtol=1e-10)
    np.testing.assert_array_equal(r0["lb"], r1["lb"])
    np.testing.assert_allclose(np.concatenate([r0["mu"], r1["mu"]], axis=1), one["mu_beta_vb"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(r0["theta"], one["theta_vb"], rtol=1e-8, atol=1e-11)
    np.testing.assert_array_equal(r0["theta"], r1["theta"])


def test_golden_fixture_on_gpu():
    """tests/golden/vb_toy_anneal.npz (oracle-produced, see make_golden.py) through the HIP path."""
    import atlasqtl_amd as A
    from tests.test_oracle import load_vb
    for name in ("vb_toy_anneal", "vb_toy_noanneal", "vb_harmonic"):
        z, lh, li, anneal = load_vb(name)
        q = z["Y"].shape[1]
        got = A.atlasqtl_global_local_core_(z["Y"], z["X"], q, anneal, 1, 0.1, 1000, 0, lh, li, full_output=True,
                                            debug=True)
        assert got["it"] == int(z["out_it"])
        np.testing.assert_allclose(got["elbo_trace"][1], z["out_elbo_lb"], rtol=1e-9)
        np.testing.assert_allclose(got["mu_beta_vb"], z["out_mu_beta_vb"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(got["gam_vb"], z["out_gam_vb"], atol=1e-9)


def test_missing_values_in_y_match_oracle():
    """NaN in Y (the reference's coreDualMisLoop branch, src/coreLoop.cpp:91-138, R/atlasqtl_global_local_core.R:19-32):
    the generic n-space kernel with a masked residual against the oracle's Gram-space restatement."""
    import atlasqtl_amd as A
    from tests.test_oracle import load_vb
    z, lh, li, anneal = load_vb("vb_toy_missing")
    assert np.isnan(z["Y"]).any()
    got = A.atlasqtl_global_local_core_(z["Y"], z["X"], z["Y"].shape[1], anneal, 1, 0.1, 1000, 0, lh, li,
                                        full_output=True, debug=True)
    assert got["it"] == int(z["out_it"]) and got["converged"]
    np.testing.assert_allclose(got["elbo_trace"][1], z["out_elbo_lb"], rtol=1e-9)
    np.testing.assert_allclose(got["mu_beta_vb"], z["out_mu_beta_vb"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(got["gam_vb"], z["out_gam_vb"], atol=1e-9)
    np.testing.assert_allclose(got["theta_vb"], z["out_theta_vb"], rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("shape,na", [((100, 75, 20), 0.0), ((300, 130, 49), 0.0), ((200, 90, 33), 0.08), ((70, 17, 1), 0.1)])
def test_generic_kernel_matches_oracle(shape, na, monkeypatch):
    """The wave-per-trait kernel (forced with AQ_KERNEL=2) on complete and on incomplete Y."""
    import atlasqtl_amd as A
    from oracle import atlasqtl_oracle as O
    from tests.util import make_problem
    monkeypatch.setenv("AQ_KERNEL", "2")
    n, p, q = shape
    prob = make_problem(n, p, q, p_act=min(8, p // 3), prob_assoc=1.0 if q <= 20 else 0.3, na_frac=na)
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, (1, 2, 10), 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"], trace=tr, full_output=True)
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, (1, 2, 10), 1, 0.1, 1000, 0, prob["list_hyper"],
                                        prob["list_init"], full_output=True, debug=True)
    assert got["it"] == ref["it"]
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(got["elbo_trace"][1], lref, rtol=1e-9)
    np.testing.assert_allclose(got["mu_beta_vb"], ref["mu_beta_vb"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], atol=1e-9)
    np.testing.assert_allclose(got["tau_vb"], ref["tau_vb"], rtol=1e-8)


def _check_against_oracle(prob, q, kernel=None):
    import atlasqtl_amd as A
    from oracle import atlasqtl_oracle as O
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, (1, 2, 10), 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"], trace=tr, full_output=True)
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, (1, 2, 10), 1, 0.1, 1000, 0, prob["list_hyper"],
                                        prob["list_init"], full_output=True, debug=True)
    if kernel is not None:
        assert got["core_kernel"] == kernel
    assert got["it"] == ref["it"]
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(got["elbo_trace"][1], lref, rtol=1e-9)
    np.testing.assert_allclose(got["mu_beta_vb"], ref["mu_beta_vb"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(got["gam_vb"], ref["gam_vb"], atol=1e-9)
    np.testing.assert_allclose(got["theta_vb"], ref["theta_vb"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(got["tau_vb"], ref["tau_vb"], rtol=1e-8)


@pytest.mark.parametrize("wpt", [2, 4])
@pytest.mark.parametrize("shape,na", [((300, 130, 49), 0.0), ((200, 90, 33), 0.08)])
def test_generic_kernel_split_traits_matches_oracle(shape, na, wpt, monkeypatch):
    """Generic kernel with a trait's samples split over 2 / 4 waves (the n > 2048 geometry, forced at small n)."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_KERNEL", "2")
    monkeypatch.setenv("AQ_TW_WPT", str(wpt))
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=8, prob_assoc=0.3, na_frac=na), q)


@pytest.mark.parametrize("force_old", [False, True])
@pytest.mark.parametrize("shape,na", [((2500, 40, 20), 0.05), ((5000, 24, 17), 0.05), ((5200, 20, 9), 0.0), ((2500, 40, 20), 0.0),
                                      ((1100, 50, 33), 0.0), ((1100, 50, 33), 0.1)])
def test_large_n_matches_oracle(shape, na, force_old, monkeypatch):
    """n beyond one workgroup's registers (C5 has n = 5000 and a missingness mask): the sample axis split over cooperating
    workgroups per trait tile.  Default: the look-ahead kernel (core_kernel 0: partial S' exchanged by the recurrence waves,
    redundant chains; MASK instances with per-trait Gram blocks from HBM when Y has missing values); AQ_KERNEL=3: the
    two-barrier masked kernel, which stays the fallback when those blocks do not fit the memory."""
    from tests.util import make_problem
    if force_old:
        monkeypatch.setenv("AQ_KERNEL", "3")
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=6, prob_assoc=0.5, na_frac=na), q, kernel=3 if force_old else 0)


@pytest.mark.parametrize("xhelper", [0, 1])
@pytest.mark.parametrize("C", [2, 3, 5, 8])
@pytest.mark.parametrize("shape", [(300, 130, 49), (200, 90, 33), (1000, 64, 17)])
def test_look_ahead_kernel_sample_split_matches_oracle(shape, C, xhelper, monkeypatch):
    """The cross-workgroup exchange of the look-ahead kernel's partial S' (agent-scope stores + flags, the same chain run
    redundantly in every part) forced at small n with AQ_LA_C; the exchange run by the recurrence wave at the start of its
    chain and by the helper wave a block ahead (the host's choice for long matrix phases)."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_LA_C", str(C))
    monkeypatch.setenv("AQ_LA_XHELPER", str(xhelper))
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=8, prob_assoc=0.3), q, kernel=0)


@pytest.mark.parametrize("shape,na", [((2500, 40, 20), 0.05), ((5000, 24, 17), 0.05)])
def test_large_n_generic_kernel_matches_oracle(shape, na, monkeypatch):
    """The generic kernel's 2 / 4 waves-per-trait geometry at its natural sizes (forced with AQ_KERNEL=2)."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_KERNEL", "2")
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=6, prob_assoc=0.5, na_frac=na), q, kernel=2)


@pytest.mark.parametrize("C", [2, 3, 5])
@pytest.mark.parametrize("shape,na", [((300, 130, 49), 0.04), ((200, 90, 33), 0.0)])
def test_masked_kernel_sample_split_matches_oracle(shape, na, C, monkeypatch):
    """The cross-workgroup exchange of partial S (release/acquire flags, redundant recursion) forced at small n."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_MIS_C", str(C))
    monkeypatch.setenv("AQ_KERNEL", "3")
    n, p, q = shape
    prob = make_problem(n, p, q, p_act=8, prob_assoc=0.3, na_frac=na)
    if na == 0.0:   # complete Y reaches the masked kernel only beyond the look-ahead kernel's n; poke one NaN in
        prob["Y"][3, 1] = np.nan
    _check_against_oracle(prob, q, kernel=3)


@pytest.mark.parametrize("shape,na", [((70, 17, 1), 0.1), ((128, 40, 16), 0.3), ((200, 90, 33), 0.08), ((300, 130, 49), 0.02),
                                      ((600, 50, 20), 0.05), ((1100, 40, 18), 0.05), ((2048, 33, 17), 0.03)])
def test_masked_mfma_kernel_matches_oracle(shape, na, monkeypatch):
    """Missing values in Y on the two-barrier masked f64-MFMA kernel (aq_core_sweep_mis.h, core_kernel == 3, forced with
    AQ_KERNEL=3: the fallback of the look-ahead MASK path): every residual-tile geometry NT = 1, 2, 4, 8, 16, ragged p and q,
    one trait, 30 % missing."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_KERNEL", "3")
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=min(6, p // 3), prob_assoc=1.0 if q <= 20 else 0.3, na_frac=na), q, kernel=3)


@pytest.mark.parametrize("env", [{}, {"AQ_CHAIN": "3"}, {"AQ_LA_C": "2"}, {"AQ_LA_C": "5"}, {"AQ_LA_C": "3", "AQ_LA_XHELPER": "1"}])
@pytest.mark.parametrize("shape,na", [((70, 17, 1), 0.1), ((128, 40, 16), 0.3), ((200, 90, 33), 0.08), ((300, 130, 49), 0.02),
                                      ((600, 50, 20), 0.05), ((1000, 64, 18), 0.05)])
def test_look_ahead_mask_kernel_matches_oracle(shape, na, env, monkeypatch):
    """Missing values in Y on the look-ahead kernel (MASK instances, core_kernel == 0): the residual re-masked after every
    update, the traits' own diagonal and cross Gram blocks precomputed into HBM and staged through LDS, per-entry
    sig2_beta_vb, the NA forms of the column sums; plain, with chained SNP segments and with the sample split."""
    from tests.util import make_problem
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=min(6, p // 3), prob_assoc=1.0 if q <= 20 else 0.3, na_frac=na), q, kernel=0)


def test_mask_path_falls_back_when_the_gram_blocks_do_not_fit(monkeypatch):
    from tests.util import make_problem
    monkeypatch.setenv("AQ_GK_MAX_GB", "0.000001")
    _check_against_oracle(make_problem(200, 90, 33, p_act=6, prob_assoc=0.3, na_frac=0.08), 33, kernel=3)


def test_missing_beyond_index_list_capacity_falls_back_to_generic():
    """More than AQ_MIS_MMAX = 1024 missing samples in a trait: the generic kernel takes over (core_kernel == 2)."""
    from tests.util import make_problem
    prob = make_problem(3000, 24, 5, p_act=4, prob_assoc=1.0, na_frac=0.4)
    assert np.isnan(prob["Y"]).sum(axis=0).max() > 1024
    _check_against_oracle(prob, 5, kernel=2)


@pytest.mark.parametrize("chain", [2, 5])
@pytest.mark.parametrize("shape,na", [((300, 130, 49), 0.04), ((200, 90, 33), 0.08)])
def test_masked_kernel_chained_segments_match_oracle(shape, na, chain, monkeypatch):
    """Chained SNP segments of the masked kernel (per-tile release/acquire hand-off of the residual, per-segment sums),
    forced at small size with AQ_CHAIN."""
    from tests.util import make_problem
    monkeypatch.setenv("AQ_CHAIN", str(chain))
    monkeypatch.setenv("AQ_KERNEL", "3")
    n, p, q = shape
    _check_against_oracle(make_problem(n, p, q, p_act=8, prob_assoc=0.3, na_frac=na), q, kernel=3)
