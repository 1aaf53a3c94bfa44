"""Generates tests/golden/*.npz.  Run here (CPU container): python tests/golden/make_golden.py

The reference cannot be executed in this image (no R; src/coreLoop.cpp needs RcppEigen, see
DESIGN.md section 3) and its tests hold no numeric fixtures, so these vectors are produced by the
CPU oracle (oracle/) on seeded synthetic inputs.  They pin the oracle against regressions and give
the GPU tests fixed targets; they are NOT reference-produced numbers ("parity unpinned").
Each file holds inputs and expected outputs only (data, no code).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import atlasqtl_oracle as O  # noqa: E402
from tests.util import make_problem, operator_inputs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def vb_case(name, n, p, q, anneal, na_frac=0.0, **kw):
    prob = make_problem(n, p, q, na_frac=na_frac, **kw)
    tr = []
    out = O.atlasqtl_global_local_core_(prob["Y"], prob["X"], q, anneal, 1, 0.1, 1000, prob["list_hyper"],
                                        prob["list_init"], trace=tr, full_output=True)
    lh, li = prob["list_hyper"], prob["list_init"]
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), X=prob["X"], Y=prob["Y"],
        anneal=np.array(anneal if anneal is not None else [0, 0, 0], dtype=np.float64),
        has_anneal=np.array(anneal is not None),
        **{"hyper_" + k: np.asarray(lh[k]) for k in ("A2_inv", "eta", "kappa", "m0", "n0", "nu", "rho", "t02")},
        **{"init_" + k: np.asarray(li[k]) for k in ("gam_vb", "mu_beta_vb", "sig02_inv_vb", "sig2_beta_vb",
                                                    "sig2_theta_vb", "tau_vb", "theta_vb", "zeta_vb")},
        out_it=np.array(out["it"]), out_converged=np.array(out["converged"]), out_lb_opt=np.array(out["lb_opt"]),
        out_elbo_it=np.array([r["it"] for r in tr if r["lb"] is not None]),
        out_elbo_lb=np.array([r["lb"] for r in tr if r["lb"] is not None]),
        out_gam_vb=out["gam_vb"], out_mu_beta_vb=out["mu_beta_vb"], out_theta_vb=out["theta_vb"],
        out_zeta_vb=out["zeta_vb"], out_tau_vb=out["tau_vb"], out_lam2_inv_vb=out["lam2_inv_vb"])
    print(name, "it", out["it"], "lb", out["lb_opt"])


def op_case(name, p, q, mis, c):
    a = operator_inputs(p, q, seed=11, mis=mis, c=c)
    inp = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.items() if k != "cp_X_rm"}
    if mis:
        O.core_dual_mis_loop(a["cp_X"], a["cp_X_rm"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"],
                             a["log_sig2_inv_vb"], a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"],
                             a["sig2_beta_vb"], a["tau_vb"], a["shuffled_ind"], a["sample_q"], c=c)
        inp["cp_X_rm"] = np.stack(a["cp_X_rm"])
    else:
        O.core_dual_loop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                         a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"], a["sig2_beta_vb"],
                         a["tau_vb"], a["shuffled_ind"], a["sample_q"], c=c)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{"in_" + k: np.asarray(v) for k, v in inp.items()},
                        out_gam_vb=a["gam_vb"], out_mu_beta_vb=a["mu_beta_vb"], out_m1_beta=a["m1_beta"],
                        out_cp_betaX_X=a["cp_betaX_X"])
    print(name, "ok")


if __name__ == "__main__":
    vb_case("vb_toy_anneal", 100, 75, 20, (1, 2, 10), p_act=10, prob_assoc=1.0)      # the reference test's shape
    vb_case("vb_toy_noanneal", 100, 75, 20, None, p_act=10, prob_assoc=1.0)
    vb_case("vb_toy_missing", 100, 60, 12, (1, 2, 10), na_frac=0.05, p_act=8, prob_assoc=1.0)
    vb_case("vb_harmonic", 90, 33, 17, (2, 3, 5), p_act=6, prob_assoc=1.0)
    op_case("op_core_dual_loop", 37, 9, False, 0.8)
    op_case("op_core_dual_mis_loop", 23, 5, True, 1.0)
