"""Developer check: error magnitudes vs the oracle and timings at larger sizes (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import make_problem
import atlasqtl_amd as A
from atlasqtl_amd.core import VbRun

def parity(n, p, q, anneal=(1, 2, 10)):
    from oracle import atlasqtl_oracle as O
    prob = make_problem(n, p, q, p_act=10, prob_assoc=0.2 if q > 20 else 1.0)
    tr = []
    t = time.time()
    ref = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, anneal, 1, 0.1, 1000, prob["list_hyper"], prob["list_init"], trace=tr, full_output=True)
    t_ref = time.time() - t
    t = time.time()
    got = A.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, anneal, 1, 0.1, 1000, 0, prob["list_hyper"], prob["list_init"], full_output=True, debug=True)
    t_gpu = time.time() - t
    its, lbs = got["elbo_trace"]
    rl = np.array([r["lb"] for r in tr if r["lb"] is not None])
    rel = lambda a, b, f=1e-8: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), f)))
    print(f"[parity n={n} p={p} q={q} TT={os.environ.get('AQ_TT','auto')}] it {got['it']} vs {ref['it']}; elbo rel {np.max(np.abs(lbs-rl)/np.abs(rl)):.2e}; "
          f"mu rel {rel(got['mu_beta_vb'], ref['mu_beta_vb']):.2e}; gam abs {np.max(np.abs(got['gam_vb']-ref['gam_vb'])):.2e}; "
          f"theta rel {rel(got['theta_vb'], ref['theta_vb'],1e-6):.2e}; oracle {t_ref:.2f}s gpu {t_gpu:.2f}s core_ms/launch {got['core_ms']/max(got['core_launches'],1):.3f}", flush=True)

def timing(n, p, q, sweeps=6, anneal=(1, 2, 10), na_frac=0.0):
    prob = make_problem(n, p, q, p_act=20, prob_assoc=0.2, maf=(0.05, 0.5), na_frac=na_frac)
    t = time.time()
    run = VbRun(prob["Y"], prob["X"], prob["list_hyper"], prob["list_init"], anneal, 1e-9, 1000)
    t_create = time.time() - t
    run.run_sweeps(1)
    st0 = run.status()
    t = time.time()
    run.run_sweeps(sweeps)
    dt = time.time() - t
    st = run.status()
    ms = (st["core_ms"] - st0["core_ms"]) / max(st["core_launches"] - st0["core_launches"], 1)
    flop = 4.0 * n * prob["p"] * q
    print(f"[timing n={n} p={prob['p']} q={q} TT={os.environ.get('AQ_TT','auto')}] create {t_create:.2f}s; {sweeps} sweeps {dt*1e3/sweeps:.2f} ms/sweep; core {ms:.3f} ms "
          f"=> {flop/ms/1e9:.2f} TFLOP/s fp64; it={st['it']} lb={st['lb_opt']}", flush=True)
    run.close()

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "parity"):
        parity(100, 75, 20)
        parity(200, 500, 50)
        parity(1000, 800, 100)
    if what in ("all", "timing"):
        timing(1000, 5000, 1000)
    if what == "tiles":
        for q in (1024, 2048, 4096, 4112, 6144, 8192, 10000, 12288):
            timing(1000, 3200, q, sweeps=3)
    if what == "generic":
        timing(1000, 5000, 1000)
        timing(1000, 5000, 1000, na_frac=0.05)
        timing(1000, 5000, 4096)
        timing(2000, 3000, 1024)
    if what == "mis":   # masked MFMA kernel vs generic kernel, same data
        timing(1000, 4000, 4096, sweeps=2, na_frac=0.05)
        timing(500, 4000, 4096, sweeps=2, na_frac=0.05)
        timing(2000, 2000, 4096, sweeps=2, na_frac=0.05)
    if what == "mischain":   # more tiles than CUs with missing values: chained segments of the masked kernel
        timing(1000, 3200, 10000, sweeps=2, na_frac=0.05)
    if what == "misbig":   # sample-split masked kernel at C5-like n
        timing(5000, 2000, 2512, sweeps=2, na_frac=0.05)
        timing(5000, 2000, 2512, sweeps=2)
        timing(2000, 2000, 4096, sweeps=2)
    if what == "bign":   # generic kernel beyond n = 2048 (C5-like n, reduced p and q)
        timing(1000, 4000, 4096, sweeps=2, na_frac=0.05)
        timing(5000, 2000, 2512, sweeps=2, na_frac=0.05)
        timing(5000, 2000, 2512, sweeps=2)
    if what == "c3":
        timing(1000, 50000, 10000, sweeps=3)
    if what == "mid":
        timing(1000, 5000, 8192, sweeps=3)
