"""Hyper-parameter and initial-value objects, mirroring the reference's
``set_hyper`` / ``set_init`` / ``auto_set_hyper_`` / ``auto_set_init_``.

Follows R/set_hyper_init.R:98-141 (set_hyper), :146-197 (auto_set_hyper_),
:311-350 (set_init), :356-418 (auto_set_init_), R/utils.R:218-242 (E_Phi_X,
E_Phi_X_2 with Owen's T, get_V_p_t, get_mu) and R/prepare_atlasqtl.R:131-248
(prepare_list_hyper_ / prepare_list_init_).  R's RNG stream cannot be
reproduced without R, so the automatic initialisation draws from the same
distributions with NumPy's Generator (seeded by ``user_seed``).
"""
from __future__ import annotations

import numpy as np
from scipy import optimize, special, stats

from .prepare import (AtlasqtlError, check_matrix_, check_natural_, check_positive_, check_vector_,
                      check_zero_one_)


class ListHyper(dict):
    """R list of class "hyper" (user-built) or "out_hyper" (automatic / already checked)."""
    cls = "hyper"


class ListInit(dict):
    """R list of class "init" (user-built) or "out_init"."""
    cls = "init"


def _rep(x, q, name):
    a = check_vector_(x, name, size=(1, q))
    return np.full(q, a[0]) if a.size == 1 else a.copy()


def set_hyper(q, p, eta, kappa, n0, nu, rho, t02):
    """R/set_hyper_init.R:98-141."""
    check_vector_(q, "q", size=1); check_natural_(q, "q")
    check_vector_(p, "p", size=1); check_natural_(p, "p")
    q = int(q); p = int(p)
    n0 = _rep(n0, q, "n0")
    check_vector_(t02, "t02", size=1); check_positive_(t02, "t02")
    check_vector_(nu, "nu", size=1); check_positive_(nu, "nu")
    check_vector_(rho, "rho", size=1); check_positive_(rho, "rho")
    check_positive_(eta, "eta"); eta = _rep(eta, q, "eta")
    check_positive_(kappa, "kappa"); kappa = _rep(kappa, q, "kappa")
    h = ListHyper(q_hyper=q, p_hyper=p, A2_inv=1.0, eta=eta, kappa=kappa, m0=0.0, n0=n0, nu=float(nu),
                  rho=float(rho), t02=float(t02))
    h.cls = "hyper"
    return h


def E_Phi_X(mu, s2):                                         # R/utils.R:218-222
    return stats.norm.cdf(mu / np.sqrt(1 + s2))


def E_Phi_X_2(mu, s2):                                       # R/utils.R:224-229 (PowerTOST::OwensT)
    return stats.norm.cdf(mu / np.sqrt(1 + s2)) - 2 * special.owens_t(mu / np.sqrt(1 + s2),
                                                                      1 / np.sqrt(1 + 2 * s2))


def get_V_p_t(mu, s2, p):                                    # R/utils.R:231-235
    return p * (p - 1) * E_Phi_X_2(mu, s2) - p ** 2 * E_Phi_X(mu, s2) ** 2 + p * E_Phi_X(mu, s2)


def get_mu(E_p_t, s2, p):                                    # R/utils.R:238-242
    return np.sqrt(1 + s2) * stats.norm.ppf(E_p_t / p)


def _solve_t02(p, p0):
    """uniroot of R/set_hyper_init.R:161-170 on [1e-6, 1e5]."""
    E_p_t, V_p_t = float(p0[0]), float(p0[1])
    f = lambda x: get_V_p_t(get_mu(E_p_t, x, p), x, p) - V_p_t  # noqa: E731
    try:
        return optimize.brentq(f, 1e-6, 1e5, xtol=1e-10, rtol=1e-12)
    except ValueError as e:
        raise AtlasqtlError("No hyperparameter values matching the expectation and variance of the number of "
                            "active predictors per responses supplied in p0.Please change p0.") from e


def _median_col_var(Y):
    v = np.nanvar(Y, axis=0, ddof=1)
    return float(np.median(v))


def auto_set_hyper_(Y, p, p0):
    """R/set_hyper_init.R:146-197."""
    q = Y.shape[1]
    with np.errstate(divide="ignore"):
        eta = 1.0 / _median_col_var(Y)
    if not np.isfinite(eta):
        eta = 1e3
    t02 = _solve_t02(p, p0)
    n0 = get_mu(float(p0[0]), t02, p)
    check_positive_(t02, "t02")
    h = ListHyper(q_hyper=q, p_hyper=p, A2_inv=1.0, eta=np.full(q, eta), kappa=np.ones(q), m0=0.0,
                  n0=np.full(q, n0), nu=1e-2, rho=1.0, t02=float(t02))
    h.cls = "out_hyper"
    return h


def set_init(q, p, gam_vb, mu_beta_vb, sig02_inv_vb, sig2_beta_vb, sig2_theta_vb, tau_vb, theta_vb, zeta_vb):
    """R/set_hyper_init.R:311-350."""
    check_vector_(q, "q", size=1); check_natural_(q, "q")
    check_vector_(p, "p", size=1); check_natural_(p, "p")
    q = int(q); p = int(p)
    gam_vb = check_matrix_(gam_vb, "gam_vb", shape=(p, q)); check_zero_one_(gam_vb, "gam_vb")
    mu_beta_vb = check_matrix_(mu_beta_vb, "mu_beta_vb", shape=(p, q))
    check_vector_(sig02_inv_vb, "sig02_inv_vb", size=1); check_positive_(sig02_inv_vb, "sig02_inv_vb")
    sig2_beta_vb = check_vector_(sig2_beta_vb, "sig2_beta_vb", size=q); check_positive_(sig2_beta_vb, "sig2_beta_vb")
    sig2_theta_vb = check_vector_(sig2_theta_vb, "sig2_theta_vb", size=p)
    check_positive_(sig2_theta_vb, "sig2_theta_vb")
    tau_vb = check_vector_(tau_vb, "tau_vb", size=q); check_positive_(tau_vb, "tau_vb")
    theta_vb = check_vector_(theta_vb, "theta_vb", size=p)
    zeta_vb = check_vector_(zeta_vb, "zeta_vb", size=q)
    li = ListInit(q_init=q, p_init=p, gam_vb=gam_vb.copy(), mu_beta_vb=mu_beta_vb.copy(),
                  sig02_inv_vb=float(np.asarray(sig02_inv_vb).reshape(-1)[0]), sig2_beta_vb=sig2_beta_vb.copy(),
                  sig2_theta_vb=sig2_theta_vb.copy(), tau_vb=tau_vb.copy(), theta_vb=theta_vb.copy(),
                  zeta_vb=zeta_vb.copy())
    li.cls = "init"
    return li


def auto_set_init_(Y, p, p0, shr_fac_inv, user_seed, device_init=False):
    """R/set_hyper_init.R:356-418 (same distributions, NumPy generator).  device_init: the two p x q matrices are not
    drawn here but on the GPU when the run is created (Philox stream keyed by the seed; gam_vb / mu_beta_vb are None and
    the list carries device_seed, device_gam_mean, device_gam_sd) -- no 2 x 8pq bytes on the host."""
    q = Y.shape[1]
    rng = np.random.default_rng(user_seed)
    t02 = _solve_t02(p, p0)
    n0 = get_mu(float(p0[0]), t02, p)
    s02 = 1e-4
    check_positive_(t02, "t02")
    if device_init:
        gam_vb = mu_beta_vb = None
        device_seed = int(rng.integers(0, 2 ** 63 - 1))
    else:
        gam_vb = stats.norm.cdf(rng.normal(loc=n0, scale=s02 + t02, size=(p, q)))    # :385  [sic: sd]
        mu_beta_vb = rng.normal(size=(p, q))                                         # :387
    sig2_inv_vb = 1e-2
    with np.errstate(divide="ignore"):
        tau = 1.0 / _median_col_var(Y)                                               # :391
    if not np.isfinite(tau):
        tau = 1e3
    tau_vb = np.full(q, tau)
    sig2_beta_vb = 1.0 / rng.gamma(shape=2.0, scale=sig2_inv_vb * tau_vb)            # rate = 1/(sig2_inv*tau)
    sig02_inv_vb = float(rng.gamma(shape=max(p, q), scale=1.0))                      # :397
    theta_vb = rng.normal(scale=1.0 / np.sqrt(sig02_inv_vb * shr_fac_inv), size=p)   # :399
    sig2_theta_vb = 1.0 / (q + rng.gamma(shape=sig02_inv_vb * shr_fac_inv, scale=1.0, size=p))   # :400
    zeta_vb = rng.normal(loc=n0, scale=np.sqrt(t02), size=q)                         # :402
    li = ListInit(q_init=q, p_init=p, gam_vb=gam_vb, mu_beta_vb=mu_beta_vb, sig02_inv_vb=sig02_inv_vb,
                  sig2_beta_vb=sig2_beta_vb, sig2_theta_vb=sig2_theta_vb, tau_vb=tau_vb, theta_vb=theta_vb,
                  zeta_vb=zeta_vb)
    li.cls = "out_init"
    if device_init:
        li["device_seed"], li["device_gam_mean"], li["device_gam_sd"] = device_seed, float(n0), float(s02 + t02)
    return li


def prepare_list_hyper_(list_hyper, Y, p, p0, bool_rmvd_x):
    """R/prepare_atlasqtl.R:131-181."""
    q = Y.shape[1]
    if list_hyper is None:
        return auto_set_hyper_(Y, p, p0)
    if not isinstance(list_hyper, ListHyper):
        raise AtlasqtlError("The provided list_hyper must be an object of class ``hyper'' or ``out_hyper''. \n "
                            "*** you must either use the function set_hyper to set your own hyperparameters or "
                            "list_hyper to NULL for automatic choice. ***")
    p_match = len(bool_rmvd_x) if list_hyper.cls == "hyper" else p
    if list_hyper["q_hyper"] != q:
        raise AtlasqtlError("The dimensions (q) of the provided hyperparameters (list_hyper) are not consistent "
                            "with that of Y.\n")
    if list_hyper["p_hyper"] != p_match:
        raise AtlasqtlError("The dimensions (p) of the provided hyperparameters (list_hyper) are not consistent "
                            "with that of X.\n")
    out = ListHyper(list_hyper)
    out.cls = "out_hyper"
    return out


def prepare_list_init_(list_init, Y, p, p0, bool_rmvd_x, shr_fac_inv, user_seed, device_init=False):
    """R/prepare_atlasqtl.R:189-248."""
    q = Y.shape[1]
    if list_init is None:
        return auto_set_init_(Y, p, p0, shr_fac_inv, user_seed, device_init=device_init)
    if not isinstance(list_init, ListInit):
        raise AtlasqtlError("The provided list_init must be an object of class ``init'' or `` out_init''. \n "
                            "*** you must either use the function set_init to set your own initialization or "
                            "set the argument list_init to NULL for automatic initialization. ***")
    p_match = len(bool_rmvd_x) if list_init.cls == "init" else p
    if list_init["q_init"] != q:
        raise AtlasqtlError("The dimensions (q) of the provided initial parameters (list_init) are not "
                            "consistent with that of Y.\n")
    if list_init["p_init"] != p_match:
        raise AtlasqtlError("The dimensions (p) of the provided initial parameters (list_init) are not "
                            "consistent with that of X.\n")
    out = ListInit(list_init)
    if list_init.cls == "init":
        keep = ~np.asarray(bool_rmvd_x)
        out["gam_vb"] = np.asarray(list_init["gam_vb"])[keep, :]
        out["mu_beta_vb"] = np.asarray(list_init["mu_beta_vb"])[keep, :]
        # the reference leaves the p-vectors untouched; drop them too when their length is the raw p
        for key in ("sig2_theta_vb", "theta_vb"):
            v = np.asarray(list_init[key])
            if v.shape[0] == len(bool_rmvd_x) and len(bool_rmvd_x) != p:
                out[key] = v[keep]
    out.cls = "out_init"
    return out
