"""ELBO-to-tolerance wall-clock: the reference's default run (tol = 0.1, maxit = 1000, anneal = c(1,2,10),
thinned ELBO schedule, monotonicity check on) from loop entry to `converged` (R/atlasqtl.R:179-182)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_problem
from atlasqtl_amd.core import VbRun

def main():
    n, p, q = (int(x) for x in sys.argv[1:4])
    maxit = int(sys.argv[4]) if len(sys.argv) > 4 else 1000       # the reference's default; C3 needs more (bench.py --to-tol-maxit)
    X, Y, lh, li = build_problem(n, p, q, 0, q, 0)
    run = VbRun(Y, X, lh, li, (1, 2, 10), tol=0.1, maxit=maxit, thinned_elbo_eval=True, debug=True, device=0, q_total=q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = run.status()
    its, lbs = run.elbo_trace()
    print(json.dumps({"workload": f"n={n} p={p} q={q}", "elbo_to_tol_s": dt, "it": st["it"], "converged": bool(st["converged"]), "maxit": maxit,
                      "lb_opt": st["lb_opt"], "n_elbo_evals": int(len(its)), "sweeps_per_s": st["it"] / dt,
                      "elbo_monotone": bool(np.all(np.diff(lbs) > -1.5e-8)), "core_ms_avg": st["core_ms"] / max(st["core_launches"], 1)}))
    run.close()

if __name__ == "__main__":
    main()
