"""Shared problem builders for the test-suite (inputs only; no oracle logic here)."""
import numpy as np

from atlasqtl_amd import hyper_init as H
from atlasqtl_amd import synth
from oracle import prepare_oracle


def make_problem(n, p, q, p_act=10, seed=123, init_seed=456, maf=0.2, p0=(5, 25), prob_assoc=0.2, na_frac=0.0,
                 q_act=None):
    """Synthetic data in the shape of the reference's example generator, pre-processed by the
    host mirror of prepare_data_, with automatic hyper-parameters and a seeded automatic init."""
    d = synth.simulate(n, p, q, p_act=p_act, q_act=q_act, seed=seed, maf=maf, prob_assoc=prob_assoc, na_frac=na_frac)
    # host arrays for both sides of a parity test: the NumPy restatement of prepare_data_ (the device-side preparation has
    # its own tests against it, tests/test_gpu_prepare.py)
    X, Y, bool_cst, bool_coll = prepare_oracle.prepare_xy(d["Y"], d["X"])
    bool_rmvd_x = bool_cst.copy()
    bool_rmvd_x[~bool_cst] = bool_coll
    pp = X.shape[1]
    lh = H.prepare_list_hyper_(None, Y, pp, p0, bool_rmvd_x)
    li = H.prepare_list_init_(None, Y, pp, p0, bool_rmvd_x, q, init_seed)
    return dict(X=X, Y=Y, list_hyper=lh, list_init=li, truth=d, n=n, p=pp, q=q)


def operator_inputs(p, q, n=60, seed=0, mis=False, c=1.0):
    """Random but well-formed inputs of coreDualLoop / coreDualMisLoop (R layout)."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, p))
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    Y = rng.normal(size=(n, q))
    gam = np.asfortranarray(rng.uniform(0.01, 0.6, size=(p, q)))
    mu = np.asfortranarray(rng.normal(size=(p, q)) * 0.3)
    m1 = np.asfortranarray(gam * mu)
    theta = rng.normal(size=p) * 0.5
    zeta = rng.normal(size=q) * 0.5 - 1.5
    from scipy.special import log_ndtr
    tz = theta[:, None] + zeta[None, :]
    lP = np.asfortranarray(log_ndtr(tz))
    l1 = np.asfortranarray(log_ndtr(-tz))
    tau = rng.uniform(0.5, 2.0, size=q)
    log_tau = np.log(tau) - 0.01
    out = dict(gam_vb=gam, mu_beta_vb=mu, m1_beta=m1, log_Phi=lP, log_1mPhi=l1, tau_vb=tau, log_tau_vb=log_tau,
               log_sig2_inv_vb=-0.3, c=c, shuffled_ind=np.arange(p, dtype=np.int32),
               sample_q=np.arange(q, dtype=np.int32), X=X, Y=Y)
    if mis:
        mis_pat = (rng.random((n, q)) > 0.1).astype(np.float64)
        Y0 = Y * mis_pat
        cp_X = np.asfortranarray(X.T @ X)
        cp_X_rm = [np.asfortranarray(X[mis_pat[:, k] == 0].T @ X[mis_pat[:, k] == 0]) for k in range(q)]
        cp_Y_X = np.asfortranarray(Y0.T @ X)
        bx = cp_X.T @ m1 - np.stack([cp_X_rm[k].T @ m1[:, k] for k in range(q)], axis=1)
        out.update(cp_X=cp_X, cp_X_rm=cp_X_rm, cp_Y_X=cp_Y_X, cp_betaX_X=np.asfortranarray(bx),
                   sig2_beta_vb=np.asfortranarray(rng.uniform(0.005, 0.02, size=(p, q))))
    else:
        cp_X = np.asfortranarray(X.T @ X)
        out.update(cp_X=cp_X, cp_Y_X=np.asfortranarray(Y.T @ X), cp_betaX_X=np.asfortranarray(cp_X.T @ m1),
                   sig2_beta_vb=rng.uniform(0.005, 0.02, size=q))
    return out
