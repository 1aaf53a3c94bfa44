"""world_size-2 gloo test (CPU): the q-sharded protocol -- each rank holds half of the traits and
the ranks exchange exactly the payload the HIP path all-reduces (p + 3 doubles per sweep, 6 on ELBO
sweeps) -- reproduces the unsharded run: same iteration count, same ELBO trace, same state."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import sharded_oracle as S
    from tests.util import make_problem
    prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)
    q = 20
    k0, k1 = (0, 16) if rank == 0 else (16, 20)       # shards are whole 16-trait tiles, as in bench.py
    lh, li = dict(prob["list_hyper"]), dict(prob["list_init"])
    for k in ("eta", "kappa", "n0"):
        lh[k] = np.asarray(lh[k])[k0:k1]
    for k in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li[k] = np.asarray(li[k])[k0:k1]
    for k in ("gam_vb", "mu_beta_vb"):
        li[k] = np.asarray(li[k])[:, k0:k1]
    calls = []

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        calls.append(t.numel())
        return t.numpy()

    tr = []
    out = S.run_sharded(prob["Y"][:, k0:k1], prob["X"], q, (1, 2, 10), 0.1, 1000, lh, li, allreduce=allreduce, trace=tr)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), it=out["it"], lb=np.array([r["lb"] for r in tr if r["lb"] is not None]),
             gam=out["gam_vb"], mu=out["mu_beta_vb"], theta=out["theta_vb"], zeta=out["zeta_vb"], calls=np.array(calls))
    dist.destroy_process_group()


def test_q_sharded_protocol_two_ranks(tmp_path):
    from oracle import atlasqtl_oracle as O
    from tests.util import make_problem
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    prob = make_problem(100, 75, 20, p_act=10, prob_assoc=1.0)
    tr = []
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], 20, (1, 2, 10), 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"], trace=tr, full_output=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["it"]) == int(r1["it"]) == ref["it"]
    lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
    np.testing.assert_allclose(r0["lb"], lref, rtol=1e-10)
    np.testing.assert_array_equal(r0["lb"], r1["lb"])            # every rank assembles the identical ELBO
    np.testing.assert_array_equal(r0["theta"], r1["theta"])      # replicated p-vector state stays identical
    gam = np.concatenate([r0["gam"], r1["gam"]], axis=1)
    mu = np.concatenate([r0["mu"], r1["mu"]], axis=1)
    np.testing.assert_allclose(gam, ref["gam_vb"], atol=1e-9)
    np.testing.assert_allclose(mu, ref["mu_beta_vb"], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(np.concatenate([r0["zeta"], r1["zeta"]]), ref["zeta_vb"], rtol=1e-8)
    # payload sizes: p + 3 per sweep (plus the initial one), 8 on ELBO sweeps
    sizes = set(int(x) for x in r0["calls"])
    assert sizes == {prob["p"] + 3, 8}
