"""aq_vb_run_multi: the whole q-sharded run from ONE host process (host threads + RCCL inside the library), the entry a host
without torch.distributed -- R, the reference's own host language -- uses for more than one GPU.
On a one-GPU box: n_gpus = 1 through the real RCCL path (a one-rank communicator: same threads, barriers, collectives and
gather as with eight), and n_gpus = 2 / 3 with every shard on device 0 and the host-staged transport -- each shard a separate
handle with its own launch plan, the payloads reduced in rank order.  All against the plain single-handle run, which the other
GPU tests hold against the oracle.  Not measured on more than one GPU (no multi-GPU box in this pool)."""
import numpy as np
import pytest

from tests.util import make_problem

pytestmark = pytest.mark.gpu


def _single(prob, anneal, maxit):
    import atlasqtl_amd as A
    return A.atlasqtl_global_local_core_(prob["Y"], prob["X"], prob["q"], anneal, 1, 0.1, maxit, 0, prob["list_hyper"],
                                         prob["list_init"], full_output=True, debug=True)


def _same(a, b, tol_mu=1e-9):
    assert a["it"] == b["it"] and a["converged"] == b["converged"]
    np.testing.assert_allclose(a["elbo_trace"][1], b["elbo_trace"][1], rtol=1e-11)
    assert list(a["elbo_trace"][0]) == list(b["elbo_trace"][0])
    for k in ("gam_vb", "mu_beta_vb", "beta_vb", "theta_vb", "zeta_vb", "tau_vb", "sig2_beta_vb", "lam2_inv_vb", "sig2_theta_vb"):
        assert np.max(np.abs(a[k] - b[k]) / np.maximum(np.abs(b[k]), 1e-6)) < tol_mu, k


def test_run_multi_one_gpu_through_rccl():
    from atlasqtl_amd.core import run_multi
    prob = make_problem(200, 500, 50, p_act=10)
    ref = _single(prob, (1, 2, 10), 1000)
    got = run_multi(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], (1, 2, 10), 0.1, 1000, n_gpus=1, transport=0)
    _same(got, ref, tol_mu=1e-13)          # one shard, the same kernels: the same bits up to the order of nothing
    assert got["seconds"] > 0 and got["core_ms"] > 0


@pytest.mark.parametrize("n_parts,na", [(2, 0.0), (3, 0.0), (3, 0.08)])
def test_run_multi_sharded_on_one_device_matches_single_run(n_parts, na):
    """Three shards of 32 / 32 / 36 traits (whole 16-trait tiles), host-staged all-reduce in rank order."""
    from atlasqtl_amd.core import run_multi, vb_partition
    prob = make_problem(150, 300, 100, p_act=12, na_frac=na)
    parts = vb_partition(prob["q"], n_parts)
    assert parts[0][0] == 0 and parts[-1][1] == prob["q"] and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    ref = _single(prob, (1, 2, 10), 400)
    got = run_multi(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], (1, 2, 10), 0.1, 400, n_gpus=n_parts,
                    devices=[0] * n_parts, transport=1)
    _same(got, ref)


def test_run_multi_device_generated_init_reproduces_single_gpu_draws():
    """init_generate: the Philox counters are (SNP, global trait), so the shards draw what the single run draws."""
    from atlasqtl_amd import hyper_init as H
    from atlasqtl_amd.core import run_multi
    prob = make_problem(120, 200, 64, p_act=8)
    li = dict(prob["list_init"])
    li.update(gam_vb=None, mu_beta_vb=None, device_seed=99, device_gam_mean=float(np.mean(prob["list_hyper"]["n0"])), device_gam_sd=0.5)
    import atlasqtl_amd as A
    ref = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], prob["q"], None, 1, 0.1, 30, 0, prob["list_hyper"], li,
                                        full_output=True, debug=True)
    got = run_multi(prob["Y"], prob["X"], prob["list_hyper"], li, None, 0.1, 30, n_gpus=2, devices=[0, 0], transport=1)
    _same(got, ref)


def test_run_multi_errors():
    from atlasqtl_amd._lib import AtlasqtlHipError
    from atlasqtl_amd.core import run_multi
    prob = make_problem(100, 75, 20, p_act=5)
    with pytest.raises(AtlasqtlHipError, match="distinct devices"):
        run_multi(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], None, 0.1, 5, n_gpus=2, devices=[0, 0], transport=0)
    with pytest.raises(AtlasqtlHipError, match="more parts than"):
        run_multi(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], None, 0.1, 5, n_gpus=3, devices=[0, 0, 0], transport=1)
    bad = dict(prob["list_hyper"]); bad["t02"] = -1.0        # a failing rank (every rank here) must not hang the others
    with pytest.raises(AtlasqtlHipError):
        run_multi(prob["Y"], prob["X"], bad, prob["list_init"], (4, 2, 10), 0.1, 5, n_gpus=2, devices=[0, 0], transport=1)
