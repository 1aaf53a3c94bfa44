"""Randomised parity sweep: random small shapes (ragged n, p, q; with and without missing values; with and without
annealing) through the HIP path against the oracle.  usage: python tests/tools/fuzz_parity.py [ncases] [seed]
(AQ_FUZZ_NMAX / AQ_FUZZ_PMAX / AQ_FUZZ_QMAX widen the shape ranges, e.g. AQ_FUZZ_NMAX=1300 crosses the sample-split boundary)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import atlasqtl_amd as A
from oracle import atlasqtl_oracle as O
from tests.util import make_problem


def run(ncases=30, seed=2024):
    """Returns the worst errors; raises AssertionError on the first mismatch."""
    rng = np.random.default_rng(seed)
    worst = dict(elbo=0.0, mu=0.0, gam=0.0)
    t0 = time.time()
    for c in range(ncases):
        n = int(rng.integers(20, int(os.environ.get("AQ_FUZZ_NMAX", 400))))
        p = int(rng.integers(12, int(os.environ.get("AQ_FUZZ_PMAX", 160))))
        q = int(rng.integers(1, int(os.environ.get("AQ_FUZZ_QMAX", 70))))
        na = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
        anneal = [None, (1, 2, 10), (2, 3, 5), (3, 2, 4)][int(rng.integers(0, 4))]
        prob = make_problem(n, p, q, p_act=max(1, min(8, p // 3)), prob_assoc=0.4, na_frac=na, seed=int(rng.integers(1, 10**6)),
                            init_seed=int(rng.integers(1, 10**6)), p0=(2, 6))
        tr = []
        args_ref = (prob["Y"], prob["X"], q, anneal, 1, 0.1, 300, prob["list_hyper"], prob["list_init"])
        args_hip = (prob["Y"], prob["X"], q, anneal, 1, 0.1, 300, 0, prob["list_hyper"], prob["list_init"])
        ref_exc = got_exc = ref = got = None
        try:
            ref = O.atlasqtl_global_local_core_(*args_ref, trace=tr, full_output=True, debug=True)
        except Exception as e:   # e.g. a non-monotone ELBO on a degenerate draw: both sides must then fail the same way
            ref_exc = e
        try:
            got = A.atlasqtl_global_local_core_(*args_hip, full_output=True, debug=True)
        except Exception as e:
            got_exc = e
        if ref_exc is not None or got_exc is not None:
            both = ref_exc is not None and got_exc is not None
            same = both and ("monotonically" in str(ref_exc)) == ("monotonically" in str(got_exc))
            assert same, f"case {c}: one-sided or different failure: oracle {ref_exc!r}, HIP {got_exc!r}"
            print(f"case {c}: n={n} p={prob['p']} q={q} na={na} anneal={anneal}: both raise ({got_exc})", flush=True)
            continue
        lref = np.array([r["lb"] for r in tr if r["lb"] is not None])
        e_elbo = float(np.max(np.abs(got["elbo_trace"][1] - lref) / np.abs(lref))) if lref.size else 0.0
        e_mu = float(np.max(np.abs(got["mu_beta_vb"] - ref["mu_beta_vb"]) / np.maximum(np.abs(ref["mu_beta_vb"]), 1e-8)))
        e_g = float(np.max(np.abs(got["gam_vb"] - ref["gam_vb"])))
        ok = got["it"] == ref["it"] and e_elbo < 1e-8 and e_mu < 1e-6 and e_g < 1e-8
        worst = dict(elbo=max(worst["elbo"], e_elbo), mu=max(worst["mu"], e_mu), gam=max(worst["gam"], e_g))
        print(f"case {c}: n={n} p={prob['p']} q={q} na={na} anneal={anneal} kernel={got['core_kernel']} it={got['it']}/{ref['it']} "
              f"elbo {e_elbo:.1e} mu {e_mu:.1e} gam {e_g:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
        assert ok, f"case {c} mismatch"
    print(f"{ncases} cases ok in {time.time() - t0:.0f} s; worst: {worst}")
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
