"""User-level entry point: ``atlasqtl()`` with the reference's argument surface
(R/atlasqtl.R:179-184) and return fields (R/atlasqtl.R:293-316,
R/atlasqtl_global_local_core.R:426-428).  Pre-processing and hyper-parameter /
initialisation handling run on the host (prepare.py, hyper_init.py); the
variational loop runs on the GPU through core.atlasqtl_global_local_core_.
"""
from __future__ import annotations

import warnings

from .core import atlasqtl_global_local_core_
from .hyper_init import prepare_list_hyper_, prepare_list_init_
from .prepare import check_annealing_, check_positive_, check_vector_, check_verbose_, prepare_data_


class AtlasqtlResult(dict):
    """The reference returns an S3 list of class "atlasqtl"; this is a dict with attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def add_collinear_back_(beta_vb, gam_vb, theta_vb, initial_colnames_X, rmvd_coll_x, names_x):
    """R/utils.R:671-733: re-insert the rows of predictors removed as duplicates, copying the
    estimates of the column each one duplicated, in the pre-removal column order."""
    pos = {nm: i for i, nm in enumerate(names_x)}
    rows = [pos[nm] if nm in pos else pos[rmvd_coll_x[nm]] for nm in initial_colnames_X]
    return beta_vb[rows, :], gam_vb[rows, :], theta_vb[rows]


def atlasqtl(Y, X, p0, anneal=(1, 2, 10), tol=0.1, maxit=1000, user_seed=None, verbose=1, list_hyper=None,
             list_init=None, save_hyper=False, save_init=False, full_output=False, thinned_elbo_eval=True,
             checkpoint_path=None, trace_path=None, add_collinear_back=False, device=0, device_init=False):
    """R/atlasqtl.R:179-322."""
    check_verbose_(verbose)
    check_annealing_(anneal)
    dat = prepare_data_(Y, X, tol, maxit, user_seed, verbose, checkpoint_path, trace_path)
    bool_rmvd_x = dat["bool_rmvd_x"]
    Xs, Yc = dat["X"], dat["Y"]
    n, p = Xs.shape
    q = Yc.shape[1]
    shr_fac_inv = q                                                         # :218
    if list_hyper is None or list_init is None:
        check_vector_(p0, "p0", size=2)
        check_positive_(p0, "p0")
    elif p0 is not None:
        warnings.warn("Provided argument p0 not used, as both list_hyper and list_init were provided.")
    list_hyper = prepare_list_hyper_(list_hyper, Yc, p, p0, bool_rmvd_x)
    list_init = prepare_list_init_(list_init, Yc, p, p0, bool_rmvd_x, shr_fac_inv, user_seed, device_init=device_init)
    if verbose != 0:
        print("**************************************************** \n"
              f"Number of samples: {n}\nNumber of (non-redundant) candidate predictors: {p}\n"
              f"Number of responses: {q}\n**************************************************** \n")
    df = 1                                                                  # :272  (hs <- TRUE, debug <- TRUE :267-268)
    res = atlasqtl_global_local_core_(Yc, Xs, shr_fac_inv, None if anneal is None else tuple(anneal), df, tol,
                                      maxit, verbose, list_hyper, list_init, checkpoint_path, trace_path,
                                      full_output, thinned_elbo_eval, debug=True, device=device)
    res = AtlasqtlResult(res)
    res["p0"] = p0
    res["rmvd_cst_x"] = dat["rmvd_cst_x"]
    res["rmvd_coll_x"] = dat["rmvd_coll_x"]
    res["names_x"], res["names_y"] = dat["names_x"], dat["names_y"]
    if add_collinear_back and dat["rmvd_coll_x"]:
        res["beta_vb"], res["gam_vb"], res["theta_vb"] = add_collinear_back_(
            res["beta_vb"], res["gam_vb"], res["theta_vb"], dat["initial_colnames_X"], dat["rmvd_coll_x"],
            dat["names_x"])
    if save_hyper:
        res["list_hyper"] = list_hyper
    if save_init:
        res["list_init"] = list_init
    return res
