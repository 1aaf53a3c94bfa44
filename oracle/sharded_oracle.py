"""oracle/sharded_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy model of the *decomposition* the HIP path uses, so that it can be checked on CPU
(and across torch.distributed ranks with the gloo backend) against the line-by-line
restatement in atlasqtl_oracle.py:

  * n-space residual R = Y - X beta instead of cp_X / cp_X_Xbeta;
    kappa_vb from ||R_k||^2 (R/update_vb.R:136-157 rewritten);
  * Z = a + gam*b, with only row/column sums kept (R/update_vb.R:217-234);
  * the trait axis sharded over ranks: every rank holds its own columns; per sweep ONE
    all-reduce of [rowSums(Z) (p), sum(gam), sum_k tau_k colSums(m2)_k, sum(zeta)], and on
    ELBO sweeps one of the 6 local ELBO sums (include/atlasqtl_hip.h, aq_vb_advance);
  * the ELBO assembled from per-trait sums (R/elbo.R:10-34 factorised).

`allreduce` is a callable(np.ndarray) -> np.ndarray summing over ranks (identity for one rank).
Parity status: as atlasqtl_oracle.py ("parity unpinned" w.r.t. reference-produced numbers).
"""
from __future__ import annotations

import numpy as np
from scipy import special as sp

from . import atlasqtl_oracle as O


def _nspace_threaded(threads, X, R, mis, xnorm, gam, lP, l1, log_sig2_inv, log_tau, m1, mu, sig2_beta, tau, c):
    """The C n-space loop (oracle_nspace_loop) over contiguous trait ranges on `threads` host threads: traits are
    independent inside the loop (src/coreLoop.cpp:58-59), every thread touches its own columns only, so the result does
    not depend on the thread count (ctypes releases the GIL during the call)."""
    q = gam.shape[1]
    threads = max(1, min(int(threads), q))
    if threads == 1:
        O.nspace_loop(X, R, mis, xnorm, gam, lP, l1, log_sig2_inv, log_tau, m1, mu, sig2_beta, tau, c=c)
        return
    from concurrent.futures import ThreadPoolExecutor
    cuts = [q * t // threads for t in range(threads + 1)]
    with ThreadPoolExecutor(threads) as ex:
        futs = [ex.submit(O.nspace_loop, X, R, mis, xnorm, gam, lP, l1, log_sig2_inv, log_tau, m1, mu, sig2_beta, tau,
                          c, cuts[t], cuts[t + 1]) for t in range(threads) if cuts[t + 1] > cuts[t]]
        for f in futs:
            f.result()


def run_sharded(Y, X, q_total, anneal, tol, maxit, list_hyper, list_init, allreduce=lambda v: v,
                thinned_elbo_eval=True, debug=True, trace=None, threads=1, return_residual=False):
    """Y, the q-vectors of list_hyper / list_init and the p x q matrices hold THIS rank's traits;
    X and the p-vectors are replicated.  Returns the same fields as the driver for the local traits.
    Y may hold NaN (missing, R/atlasqtl_global_local_core.R:19-32): the n-space form of coreDualMisLoop
    (src/coreLoop.cpp:91-138) is a masked residual with X_norm_sq = (X^2)' mis_pat in place of n - 1
    (oracle_nspace_loop with a mask), so no p x p matrix per trait is needed and the check scales to
    BASELINE-size p.  `threads`: host threads over the traits of the C loop (same results for any count)."""
    Y = np.array(Y, dtype=np.float64, order="F")
    X = np.asfortranarray(X, dtype=np.float64)
    n, p = X.shape
    q = Y.shape[1]
    if np.isnan(Y).any():                                       # :19-23
        mis = np.asfortranarray(np.where(np.isnan(Y), 0.0, 1.0))
        Y[np.isnan(Y)] = 0.0
        XN = np.asfortranarray((X ** 2).T @ mis)                # X_norm_sq, p x q
        nobs = mis.sum(axis=0)
    else:
        mis = XN = None
        nobs = np.full(q, float(n))
    shr = float(q_total)
    eta, kappa, n0 = (np.asarray(list_hyper[k], dtype=np.float64) for k in ("eta", "kappa", "n0"))
    A2_inv, m0, nu, rho, t02 = (float(list_hyper[k]) for k in ("A2_inv", "m0", "nu", "rho", "t02"))
    gam = np.array(list_init["gam_vb"], dtype=np.float64, order="F")
    mu = np.array(list_init["mu_beta_vb"], dtype=np.float64, order="F")
    sig02_inv = float(list_init["sig02_inv_vb"])
    sig2_beta = np.array(list_init["sig2_beta_vb"], dtype=np.float64)
    sig2_theta = np.array(list_init["sig2_theta_vb"], dtype=np.float64)
    tau = np.array(list_init["tau_vb"], dtype=np.float64)
    theta = np.array(list_init["theta_vb"], dtype=np.float64)
    zeta = np.array(list_init["zeta_vb"], dtype=np.float64)

    if anneal is None:
        annealing, c, it_init, ladder = False, 1.0, 1, None
    else:
        annealing = True
        ladder = O.get_annealing_ladder_(anneal)
        c, it_init = float(ladder[0]), int(anneal[2])
    c_s = c
    eps = np.finfo(np.float64).eps ** 0.5
    times_conv = np.array([1, 5, 10, 50.0]) if thinned_elbo_eval else np.array([1.0])
    batch_sched = [1, 10, 25, 50] if thinned_elbo_eval else [1]
    ind_batch_conv, batch_conv = len(batch_sched) + 1, 1
    t02_inv = 1 / t02
    sig2_zeta = 1 / (c * (p + t02_inv))
    vsld = -q_total * (np.log(t02) + np.log(p + t02_inv))
    xnorm = (X ** 2).sum(axis=0)

    def col_sums(s2b):
        """sum gam, sum m2, the X_norm_sq-weighted sum of m2 - beta^2 (R/update_vb.R:136-157 regrouped: complete Y has
        X_norm_sq = n - 1 for every entry), sum gam (log sig2_beta + 1) of R/elbo.R:27-33"""
        be = gam * mu
        s2m = s2b[None, :] if s2b.ndim == 1 else s2b
        m2 = (mu ** 2 + s2m) * gam
        w = (n - 1.0) if XN is None else XN
        return gam.sum(0), m2.sum(0), (w * (m2 - be ** 2)).sum(0), (gam * (np.log(s2m) + 1)).sum(0)

    beta = gam * mu
    R = np.asfortranarray(Y - X @ beta)
    if mis is not None:
        R *= mis
    sg, sm2, sxm, _ = col_sums(sig2_beta)
    rn = (R ** 2).sum(0)
    red = allreduce(np.concatenate([np.zeros(p), [sg.sum(), np.dot(tau, sm2), zeta.sum()]]))
    S_gam, T2 = red[p], red[p + 1]

    converged, lb_new, lb_old, it = False, -np.inf, -np.inf, 0
    nm1 = n - 1.0
    while (not converged) and it < maxit:
        lb_old = lb_new
        it += 1
        # ---- S1-S8 (aq_k_qpre)
        nu_vb = c * (nu + S_gam / 2) - c + 1
        rho_vb = c * (rho + T2 / 2)
        sig2_inv = nu_vb / rho_vb
        log_sig2_inv = sp.digamma(nu_vb) - np.log(rho_vb)
        eta_vb = c * (eta + nobs / 2 + sg / 2) - c + 1
        kappa_vb = c * (kappa + (rn + sig2_inv * sm2 + sxm) / 2)
        tau = eta_vb / kappa_vb
        if XN is None:
            sig2_beta = 1 / (c * (nm1 + sig2_inv) * tau)
        else:
            sig2_beta = np.asfortranarray(1 / (c * (XN + sig2_inv) * tau[None, :]))     # R/update_vb.R:45
        log_tau = sp.digamma(eta_vb) - np.log(kappa_vb)
        # ---- pre-pass (aq_k_prepass)
        u = theta[:, None] + zeta[None, :]
        lP, l1 = sp.log_ndtr(u), sp.log_ndtr(-u)
        if O.all_equal_1(c):
            sc, U, lPc, l1c = 1.0, u, lP, l1
        else:
            sc = np.sqrt(c); U = sc * u; lPc, l1c = sp.log_ndtr(U), sp.log_ndtr(-U)
        imr1 = O.inv_mills_ratio_(1, U, l1c, lPc)
        imr0 = O.inv_mills_ratio_(0, U, l1c, lPc)
        a_z, b_z = u + imr0 / sc, (imr1 - imr0) / sc
        # ---- core sweep in n-space (aq_core_sweep_kernel): the oracle's C port
        m1 = np.asfortranarray(gam * mu)
        _nspace_threaded(threads, X, R, mis, xnorm if XN is None else XN, gam, np.asfortranarray(lP),
                         np.asfortranarray(l1), log_sig2_inv, log_tau, m1, mu, sig2_beta, tau, c)
        sg, sm2, sxm, sgl = col_sums(sig2_beta)
        rn = (R ** 2).sum(0)
        rsZ_loc = a_z.sum(1) + (gam * b_z).sum(1)
        csZ = a_z.sum(0) + (gam * b_z).sum(0)
        # ---- the one exchange of the sweep
        red = allreduce(np.concatenate([rsZ_loc, [sg.sum(), np.dot(tau, sm2), zeta.sum()]]))
        rsZ, S_gam, T2, sum_zeta_old = red[:p], red[p], red[p + 1], red[p + 2]
        # ---- S12-S18 (replicated p-vector work)
        L = c_s * sig02_inv * shr * (theta ** 2 + sig2_theta - 2 * theta * m0 + m0 ** 2) / 2
        rho_xi_inv = c_s * (A2_inv + sig02_inv)
        if annealing:
            lam = O.update_annealed_lam2_inv_vb_(L, c_s, 1)
            Q = None
        else:
            Q = O.Q_approx_vec(L)
            lam = 1 / (Q * L) - 1
        xi_inv = 1 / rho_xi_inv
        s02 = sig02_inv * lam * shr
        sig2_theta = 1 / (c * (q_total + s02))
        theta = c * sig2_theta * (rsZ + s02 * m0 - sum_zeta_old)
        nu_s0 = c_s * (0.5 + p / 2) - c_s + 1
        rho_s0 = c_s * (xi_inv + np.sum(lam * shr * (theta ** 2 + sig2_theta - 2 * theta * m0 + m0 ** 2)) / 2)
        sig02_inv = nu_s0 / rho_s0
        # ---- S19
        zeta = c * sig2_zeta * (csZ + t02_inv * n0 - theta.sum())
        rec = dict(it=it, c=c, lb=None)
        if annealing:
            sig2_zeta = c * sig2_zeta
            c = float(ladder[it]) if it < len(ladder) else 1.0
            c_s = c
            sig2_zeta = sig2_zeta / c
            if O.all_equal_1(c):
                annealing = False
        elif it <= it_init + 1 or it % batch_conv == 0 or it % batch_conv == 1:
            # ---- ELBO from per-trait sums (aq_k_elbo_q / aq_k_elbo_final)
            eps75 = np.finfo(np.float64).eps ** 0.75
            un = theta[:, None] + zeta[None, :]
            lPn, l1n = sp.log_ndtr(un), sp.log_ndtr(-un)
            H = np.sum(gam * lPn + (1 - gam) * l1n - gam * np.log(gam + eps75) - (1 - gam) * np.log(1 - gam + eps75))
            eta_e = eta + nobs / 2 + sg / 2
            kappa_e = kappa + (rn + sig2_inv * sm2 + sxm) / 2
            log_tau_e = sp.digamma(eta_e) - np.log(kappa_e)
            loc = np.array([H, np.dot(log_tau_e, sg), sgl.sum(),
                            np.sum(nobs * (log_tau_e - np.log(2 * np.pi)) / 2 - tau * (kappa_e - sm2 * sig2_inv / 2 - kappa)),
                            np.sum((eta - eta_e) * log_tau_e - (kappa - kappa_e) * tau + eta * np.log(kappa)
                                   - eta_e * np.log(kappa_e) - sp.gammaln(eta) + sp.gammaln(eta_e)),
                            np.sum((zeta - n0) ** 2), 0.0, 0.0])
            er = allreduce(loc)
            nu_e, rho_e = nu + S_gam / 2, rho + T2 / 2
            lsi_e = sp.digamma(nu_e) - np.log(rho_e)
            l_s02 = sp.digamma(nu_s0) - np.log(rho_s0)
            l_xi = sp.digamma(1.0) - np.log(rho_xi_inv)
            A_ = er[3]
            B_ = (lsi_e * S_gam / 2 + er[1] / 2 - sig2_inv * T2 / 2 + er[0] - p * q_total * sig2_zeta / 2
                  - q_total * sig2_theta.sum() / 2 + er[2] / 2)
            C_ = O.e_theta_hs_(lam, L, l_s02 + np.log(shr), m0, theta, Q, sig02_inv * shr, sig2_theta, 1)
            D_ = (vsld - t02_inv * er[5] - q_total * t02_inv * sig2_zeta + q_total) / 2
            E_ = er[4]
            F_ = O.e_sig2_inv_hs_(xi_inv, nu_s0, l_xi, l_s02, rho_s0, sig02_inv)
            G_ = O.e_sig2_inv_(0.5, 1.0, l_xi, A2_inv, rho_xi_inv, xi_inv)
            H_ = O.e_sig2_inv_(nu, nu_e, lsi_e, rho, rho_e, sig2_inv)
            lb_new = float(A_ + B_ + C_ + D_ + E_ + F_ + G_ + H_)
            rec["lb"] = lb_new
            if debug and lb_new + eps < lb_old:
                raise O.ElboNotMonotone(f"ELBO not increasing monotonically. Exit. (it={it})")
            diff = abs(lb_new - lb_old)
            sum_exceed = int(np.sum(diff > times_conv * tol))
            if sum_exceed == 0:
                converged = True
            elif ind_batch_conv > sum_exceed:
                ind_batch_conv = sum_exceed
                batch_conv = batch_sched[sum_exceed - 1]
        if trace is not None:
            trace.append(rec)
    out = dict(beta_vb=gam * mu, gam_vb=gam, mu_beta_vb=mu, theta_vb=theta, zeta_vb=zeta, tau_vb=tau,
               converged=converged, it=it, lb_opt=lb_new, diff_lb=abs(lb_new - lb_old), lam2_inv_vb=lam)
    if return_residual:
        out["residual"] = R          # mis_pat .* (Y - X beta_vb), carried incrementally through every sweep
    return out
