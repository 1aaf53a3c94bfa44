"""Oracle parity at BASELINE-size p.  The other GPU parity tests stop at p = 500 (31 SNP blocks per chained segment); the
bench regime is 3125 blocks (781 per segment) and C5 12 500.  Here the HIP path meets the CPU oracle where the oracle can
still follow, on the host cores of the GPU box (the C loops run on one thread per group of traits: traits are independent
inside the loop, src/coreLoop.cpp:58-59, so the thread count does not change a bit of the result):

  (a) n = 1000, p = 5000, q = 48: a WHOLE annealed run to convergence against the Gram-space driver
      (oracle.atlasqtl_oracle.atlasqtl_global_local_core_: X'X is 200 MB, the reference's own formulation and update order,
      src/coreLoop.cpp:58-84 + R/atlasqtl_global_local_core.R:125-386);
  (b) C2 at full size (n = 1000, p = 5000, q = 1000): 12 sweeps = the whole ladder + 2 ELBO evaluations, against the
      n-space oracle (oracle.sharded_oracle.run_sharded);
  (c) a C3 slice (n = 1000, p = 50 000, q = 48) with the host's own launch plan and with the plan of the bench instance
      (two trait tiles per workgroup, four chained SNP segments: 781 blocks per segment) and of a multi-GPU trait shard (one
      tile per workgroup, chained);
  (d) a C5-shaped slice (n = 5000, p = 20 000, q = 32, 5 % of Y missing): MASK instances + sample split.

Besides the north-star tolerances (same `it`, ELBO <= 1e-5 relative, mu_beta_vb <= 1e-6 relative, gam_vb <= 1e-8) every case
checks the DRIFT of the incrementally updated residual: R in the handle (thousands of rank-16 updates per sweep) against
mis_pat .* (Y - X beta_vb) recomputed from the returned beta_vb."""
import os

import numpy as np
import pytest

from tests.util import make_problem

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))
_cache = {}


def _relerr(a, b, floor=1e-8):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _gpu_run(prob, anneal, maxit, tol=0.1):
    from atlasqtl_amd.core import VbRun
    run = VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], anneal, tol, maxit, True, True)
    try:
        run.run()
        st = run.status()
        res = run.result(full_output=True)
        res["residual"] = run.residual()
        res["elbo_trace"] = run.elbo_trace()
        res.update(it=st["it"], converged=bool(st["converged"]), status=st)
        return res
    finally:
        run.close()


def _check(ref, got, ref_trace, prob):
    assert got["it"] == ref["it"] and got["converged"] == ref["converged"]
    lref = np.array([r["lb"] for r in ref_trace if r["lb"] is not None])
    its, lbs = got["elbo_trace"]
    assert list(its) == [r["it"] for r in ref_trace if r["lb"] is not None]
    assert np.max(np.abs(lbs - lref) / np.abs(lref)) < 1e-5            # north star; observed ~1e-13
    assert np.max(np.abs(lbs - lref) / np.abs(lref)) < 1e-9            # what the other parity tests hold the path to
    assert np.all(np.diff(lbs) > -np.sqrt(np.finfo(float).eps))
    assert _relerr(got["mu_beta_vb"], ref["mu_beta_vb"], floor=1e-8) < 1e-6
    assert np.max(np.abs(got["gam_vb"] - ref["gam_vb"])) < 1e-8
    assert _relerr(got["theta_vb"], ref["theta_vb"], floor=1e-6) < 1e-6
    assert _relerr(got["zeta_vb"], ref["zeta_vb"], floor=1e-6) < 1e-6
    assert _relerr(got["tau_vb"], ref["tau_vb"]) < 1e-8
    # drift of the residual carried through every sweep in registers / HBM against a fresh Y - X beta
    Y = np.array(prob["Y"], dtype=np.float64)
    obs = ~np.isnan(Y)
    Y[~obs] = 0.0
    fresh = obs * (Y - prob["X"] @ got["beta_vb"])
    scale = np.sqrt((Y ** 2).sum(0) / np.maximum(obs.sum(0), 1))      # rms of each trait
    assert np.max(np.abs(got["residual"] - fresh) / scale[None, :]) < 1e-10


def test_whole_run_p5000_vs_gram_space_oracle():
    """(a): 313 SNP blocks, whole annealed run to convergence (136 sweeps on this seed), the reference's Gram-space loop."""
    from oracle import atlasqtl_oracle as O
    prob = make_problem(1000, 5000, 48, p_act=40, prob_assoc=0.3)
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], prob["q"], (1, 2, 10), 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"], trace=tr, full_output=True, threads=THREADS)
    got = _gpu_run(prob, (1, 2, 10), 1000)
    assert ref["converged"] and ref["it"] > 50
    _check(ref, got, tr, prob)
    # gam_vb ordering: stable sort, entries closer than 1e-9 count as ties
    order = np.argsort(-ref["gam_vb"].ravel(), kind="stable")[:500]
    assert np.all(np.diff(got["gam_vb"].ravel()[order]) <= 1e-9)


def _nspace_ref(key, shape, na_frac, maxit, **kw):
    if key not in _cache:
        from oracle import sharded_oracle as S
        n, p, q = shape
        prob = make_problem(n, p, q, na_frac=na_frac, **kw)
        tr = []
        ref = S.run_sharded(prob["Y"], prob["X"], prob["q"], (1, 2, 10), 0.1, maxit, prob["list_hyper"], prob["list_init"],
                            trace=tr, threads=THREADS)
        _cache[key] = (prob, ref, tr)
    return _cache[key]


def test_c2_full_size_vs_nspace_oracle():
    """(b): BASELINE config C2 at full size, the whole ladder and two ELBO evaluations."""
    prob, ref, tr = _nspace_ref("c2", (1000, 5000, 1000), 0.0, 12, p_act=40, prob_assoc=0.05)
    got = _gpu_run(prob, (1, 2, 10), 12)
    assert got["status"]["core_kernel"] == 0
    _check(ref, got, tr, prob)
    _cache.pop("c2")


@pytest.mark.parametrize("plan", ["host", "bench_instance", "trait_shard"])
def test_c3_slice_vs_nspace_oracle(plan, monkeypatch):
    """(c): full C3 p (3125 SNP blocks) on a slice of 48 traits.  `bench_instance` = the kernel instance and launch form of
    bench.py's C3 run -- <10, 9, SEG, 2>: two trait tiles per workgroup, 6 residual tiles on the recurrence wave, 13 chained SNP
    segments of 241 blocks (the host's cost model at 313 groups on 256 CUs) --, `trait_shard` = one tile per workgroup, 13 segments, as
    the q/2 shard of a two-GPU run launches it; `host` = whatever the library picks for 3 trait tiles (a sample split)."""
    nseg = 13
    if plan == "bench_instance":
        monkeypatch.setenv("AQ_TT", "2")
        monkeypatch.setenv("AQ_CHAIN", str(nseg))
    elif plan == "trait_shard":
        monkeypatch.setenv("AQ_TT", "1")
        monkeypatch.setenv("AQ_LA_NOSPLIT", "1")
        monkeypatch.setenv("AQ_CHAIN", str(nseg))
    prob, ref, tr = _nspace_ref("c3", (1000, 50000, 48), 0.0, 12, p_act=60, prob_assoc=0.3)
    got = _gpu_run(prob, (1, 2, 10), 12)
    st = got["status"]
    assert st["core_kernel"] == 0
    if plan == "bench_instance":
        assert st["tiles_per_group"] == 2 and st["chain_segments"] == nseg
    elif plan == "trait_shard":
        assert st["tiles_per_group"] == 1 and st["chain_segments"] == nseg and st["split_parts"] == 1
    _check(ref, got, tr, prob)


@pytest.mark.parametrize("plan", ["host", "chained"])
def test_c5_shaped_slice_with_missing_y_vs_nspace_oracle(plan, monkeypatch):
    """(d): n = 5000 (sample split: 313 residual tiles over several workgroups), p = 20 000 (1250 SNP blocks), 5 % of Y
    missing (MASK instances: re-masked residual, the traits' own Gram blocks streamed by LDS-DMA) against the masked n-space
    oracle (src/coreLoop.cpp:91-138 in n-space).  n = 1000 with the same mask runs the chained MASK instance C3 + NA uses."""
    if plan == "host":
        prob, ref, tr = _nspace_ref("c5", (5000, 20000, 32), 0.05, 12, p_act=40, prob_assoc=0.3)
    else:
        monkeypatch.setenv("AQ_CHAIN", "4")
        prob, ref, tr = _nspace_ref("c3na", (1000, 20000, 32), 0.05, 12, p_act=40, prob_assoc=0.3)
    got = _gpu_run(prob, (1, 2, 10), 12)
    st = got["status"]
    assert st["core_kernel"] == 0
    if plan == "host":
        assert st["split_parts"] > 1
    else:
        assert st["chain_segments"] == 4
    _check(ref, got, tr, prob)
    _cache.pop("c5" if plan == "host" else "c3na")
