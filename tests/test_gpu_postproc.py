"""GPU: post-processing on the device against the oracle's restatement of R/summarise_output.R."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (130, 49), (2000, 300)])
def test_assign_bfdr_matches_oracle(shape, idx64, monkeypatch):
    """(idx64: the 64-bit position path that 2^32 entries and more take -- C5's p q on one GPU is 4e9 -- forced at test size)"""
    import atlasqtl_amd as A
    if idx64:
        monkeypatch.setenv("AQ_BFDR_IDX64", "1")
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


def _fdr_shard_worker(rank, world, port, outdir, ties):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atlasqtl_amd.core import VbRun
    prob, gam, cuts = _fdr_problem(ties)
    q = gam.shape[1]
    k0, k1 = cuts[rank], cuts[rank + 1]
    lh, li = dict(prob["list_hyper"]), dict(prob["list_init"])
    for k in ("eta", "kappa", "n0"):
        lh[k] = np.asarray(lh[k])[k0:k1]
    for k in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li[k] = np.asarray(li[k])[k0:k1]
    li["gam_vb"] = np.asfortranarray(gam[:, k0:k1])
    li["mu_beta_vb"] = np.asfortranarray(np.asarray(li["mu_beta_vb"])[:, k0:k1])
    run = VbRun(prob["Y"][:, k0:k1], prob["X"], lh, li, None, 0.1, 5, True, False, q_total=q, process_group=dist.group.WORLD,
                trait_offset=k0)
    run.run_sweeps(0)                       # the PPIs resident on the device are the crafted initial values
    out = {}
    for thres in (0.002, 0.02, 0.05, 0.2, 0.6):
        rs, nb = run.hotspot_sizes(thres, fdr_adjust=True)
        out[f"rs_{thres}"] = rs
        out[f"nb_{thres}"] = nb
    rs, nb = run.hotspot_sizes(0.5, fdr_adjust=False)
    out["rs_plain"], out["nb_plain"] = rs, nb
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    run.close()
    dist.destroy_process_group()


def _fdr_problem(ties):
    from tests.util import make_problem
    prob = make_problem(100, 60, 50, p_act=6, prob_assoc=0.5)
    rng = np.random.default_rng(11)
    p, q = prob["p"], 50
    if ties:   # a handful of distinct values: every cutoff falls inside a big tie block that spans the ranks
        gam = rng.choice([0.9995, 0.99, 0.9, 0.5, 0.01, 1e-4], size=(p, q), p=[0.02, 0.03, 0.05, 0.1, 0.3, 0.5])
    else:
        gam = rng.beta(0.05, 1.0, size=(p, q))
        gam[rng.random((p, q)) < 0.05] = 0.97
    return prob, gam, [0, 16, 32, 50]


@pytest.mark.parametrize("ties", [False, True])
def test_fdr_hotspot_sizes_over_trait_shards_equal_the_global_ranking(ties, tmp_path):
    """assign_bFDR ranks all p q PPIs (R/summarise_output.R:207-223): three trait shards (three processes on the one GPU,
    gloo) find the end of the {FDR < thres} prefix by bisection over the PPI value and split the straddling tie block in
    original order -- same per-predictor counts as the oracle on the whole matrix."""
    import socket
    import torch.multiprocessing as mp
    from oracle import atlasqtl_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_fdr_shard_worker, args=(3, port, str(tmp_path), ties), nprocs=3, join=True)
    _, gam, _ = _fdr_problem(ties)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(3)]
    some = 0
    for thres in (0.002, 0.02, 0.05, 0.2, 0.6):
        rs_ref, nb_ref = O.hotspot_sizes(gam, thres, True)
        for r in res:
            np.testing.assert_array_equal(r[f"rs_{thres}"], rs_ref)
            assert int(r[f"nb_{thres}"]) == nb_ref
        some += nb_ref
    assert some > 0
    rs_ref, nb_ref = O.hotspot_sizes(gam, 0.5, False)
    np.testing.assert_array_equal(res[0]["rs_plain"], rs_ref)
    assert int(res[2]["nb_plain"]) == nb_ref
