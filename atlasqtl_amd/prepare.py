"""Host-side pre-processing, mirroring the reference's R checks and data preparation.

Follows R/prepare_atlasqtl.R:8-124 (``prepare_data_``, ``check_verbose_``,
``check_annealing_``) and R/utils.R:10-100, 276-343 (``check_*_``,
``rm_constant_``, ``rm_collinear_``).  Error messages keep the reference's
wording so tests read like the reference's own.  This is O(n p) host work that
runs once per call; it is not part of the accelerated sweep.
"""
from __future__ import annotations

import os

import numpy as np

_EPS75 = np.finfo(np.float64).eps ** 0.75


class AtlasqtlError(ValueError):
    """Raised where the reference calls ``stop()``."""


def check_natural_(x, name, eps=_EPS75):                      # R/utils.R:10-15
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if np.any((x < eps) | (np.abs(x - np.round(x)) > eps)):
        raise AtlasqtlError(f"{name} must be natural.")


def check_positive_(x, name, eps=_EPS75):                     # R/utils.R:17-24
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if np.any(x < eps):
        msg = f"{name} must be positive, greater than {eps:.3g}."
        if x.size > 1:
            msg = "All entries of " + msg
        raise AtlasqtlError(msg)


def check_zero_one_(x, name):                                 # R/utils.R:26-32
    x = np.asarray(x)
    if np.any(x < 0) or np.any(x > 1):
        msg = f"{name} must lie between 0 and 1."
        if x.size > 1:
            msg = "All entries of " + msg
        raise AtlasqtlError(msg)


def check_vector_(x, name, size=None, null_ok=False, na_ok=False):
    """check_structure_(x, "vector", ...) of R/utils.R:34-100 for numeric vectors."""
    if x is None:
        if null_ok:
            return None
        raise AtlasqtlError(f"{name} must be a non-empty a numeric vector.")
    a = np.atleast_1d(np.asarray(x, dtype=np.float64))
    ok = a.ndim == 1 and a.size > 0
    if size is not None:
        sizes = size if isinstance(size, (tuple, list)) else (size,)
        ok = ok and a.size in sizes
    if not na_ok:
        ok = ok and not np.any(np.isnan(a))
    ok = ok and bool(np.all(np.isfinite(a[~np.isnan(a)])))
    if not ok:
        raise AtlasqtlError(f"{name} must be a non-empty a numeric vector"
                            + (f" of length {size}" if size is not None else "")
                            + ", finite" + ("" if na_ok else " without missing value")
                            + (" or must be NULL" if null_ok else "") + ".")
    return a


def check_matrix_(x, name, shape=None, na_ok=False):
    """check_structure_(x, "matrix", ...) of R/utils.R:34-100."""
    a = np.asarray(x, dtype=np.float64)
    ok = a.ndim == 2 and a.size > 0
    if shape is not None:
        ok = ok and tuple(a.shape) == tuple(shape)
    if not na_ok:
        ok = ok and not np.any(np.isnan(a))
    ok = ok and bool(np.all(np.isfinite(a[~np.isnan(a)])))
    if not ok:
        raise AtlasqtlError(f"{name} must be a non-empty a numeric matrix"
                            + (f" of dimension {shape[0]} x {shape[1]}" if shape is not None else "")
                            + ", finite" + ("" if na_ok else " without missing value") + ".")
    return a


def check_verbose_(verbose):                                  # R/prepare_atlasqtl.R:90-95
    if verbose not in (0, 1, 2):
        raise AtlasqtlError("The verbose argument must be set to 0, 1 or 2.")


def check_annealing_(anneal):                                 # R/prepare_atlasqtl.R:100-124
    if anneal is None:
        return
    a = check_vector_(anneal, "anneal", size=3)
    check_natural_(a[[0, 2]], "anneal[c(1, 3)]")
    check_positive_(a[1], "anneal[2]")
    if a[0] not in (1, 2, 3):
        raise AtlasqtlError("The annealing spacing scheme must be set to 1 for geometric 2 for harmonic or 3 "
                            "for linear spacing.")
    if a[1] < 1.5:
        raise AtlasqtlError("Initial annealing temperature very small. May not be large enough for a "
                            "successful exploration. Please increase it or select no annealing.")
    if a[2] > 1000:
        raise AtlasqtlError("Temperature grid size very large. This may be unnecessarily computationally "
                            "demanding. Please decrease it.")


def scale_columns(X):
    """R's scale(X): centre, divide by the n-1 standard deviation.  Constant
    columns become NaN (0/0) exactly as in R, which rm_constant_ then detects."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    mean = X.mean(axis=0)
    mean = mean + (X - mean).mean(axis=0)          # second pass: R's colMeans accumulates in long double
    # a constant column has mean == its value exactly in R (0/0 = NaN below); numpy's pairwise sum can be an ulp off
    # for non-dyadic values (0.1, 1/3), which would leave a finite +-0.99 column behind
    const = X.max(axis=0) == X.min(axis=0)
    mean = np.where(const, X[0], mean)
    Xc = X - mean
    sd = np.sqrt((Xc ** 2).sum(axis=0) / (n - 1))
    with np.errstate(invalid="ignore", divide="ignore"):
        return Xc / sd


def rm_constant_(mat, names):                                 # R/utils.R:276-302
    bool_cst = np.isnan(mat.sum(axis=0))
    rmvd = [names[i] for i in np.where(bool_cst)[0]] if bool_cst.any() else None
    return mat[:, ~bool_cst], bool_cst, rmvd


def rm_collinear_(mat, names):                                # R/utils.R:304-343
    """duplicated(mat, MARGIN = 2): flag every column identical to an earlier one."""
    seen = {}
    bool_coll = np.zeros(mat.shape[1], dtype=bool)
    rmvd = {}
    for j in range(mat.shape[1]):
        key = mat[:, j].tobytes()
        if key in seen:
            bool_coll[j] = True
            rmvd[names[j]] = names[seen[key]]   # removed name -> kept name
        else:
            seen[key] = j
    return mat[:, ~bool_coll], bool_coll, (rmvd if rmvd else None)


def prepare_data_(Y, X, tol, maxit, user_seed, verbose, checkpoint_path, trace_path,
                  names_x=None, names_y=None):
    """R/prepare_atlasqtl.R:8-87.  Returns dict(Y, X, bool_rmvd_x, initial_colnames_X,
    rmvd_cst_x, rmvd_coll_x, names_x, names_y)."""
    check_vector_(user_seed, "user_seed", size=1, null_ok=True)
    check_vector_(tol, "tol", size=1)
    check_positive_(tol, "tol", eps=np.finfo(np.float64).eps)
    check_vector_(maxit, "maxit", size=1)
    check_natural_(maxit, "maxit")
    X = check_matrix_(X, "X")
    if checkpoint_path is not None and not os.path.isdir(checkpoint_path):
        raise AtlasqtlError("The directory specified in checkpoint_path does not exist. Please make sure to "
                            "provide a valid path.")
    if trace_path is not None and not os.path.isdir(trace_path):
        raise AtlasqtlError("The directory specified in trace_path does not exist. Please make sure to "
                            "provide a valid path.")
    n, p = X.shape
    Y = check_matrix_(Y, "Y", na_ok=True)
    q = Y.shape[1]
    if Y.shape[0] != n:
        raise AtlasqtlError("X and Y must have the same number of samples.")
    if np.sum(~np.isnan(Y)) / (n * q) < 0.05:
        raise AtlasqtlError("Too few non-NA values in matrix Y. Exit.")
    ind_low = (np.sum(~np.isnan(Y), axis=0) / n) < 0.025
    if ind_low.any():
        raise AtlasqtlError(f"Column(s) {list(np.where(ind_low)[0] + 1)} of matrix Y have more than 97.5% "
                            "missing values, and should be removed. Exit.")
    if names_x is None:
        names_x = [f"Cov_x_{j + 1}" for j in range(p)]
    if names_y is None:
        names_y = [f"Resp_{k + 1}" for k in range(q)]

    Xs = scale_columns(X)
    Xs, bool_cst_x, rmvd_cst_x = rm_constant_(Xs, names_x)
    names_after_cst = [nm for nm, b in zip(names_x, bool_cst_x) if not b]
    Xs, bool_coll_x, rmvd_coll_x = rm_collinear_(Xs, names_after_cst)
    bool_rmvd_x = bool_cst_x.copy()
    bool_rmvd_x[~bool_cst_x] = bool_coll_x
    if Xs.shape[1] < 1:
        raise AtlasqtlError("There must be at least 1 non-constant candidate predictor stored in X.")
    Yc = Y - np.nanmean(Y, axis=0)                         # scale(Y, center = TRUE, scale = FALSE)
    return dict(Y=Yc, X=Xs, bool_rmvd_x=bool_rmvd_x, initial_colnames_X=names_after_cst,
                rmvd_cst_x=rmvd_cst_x, rmvd_coll_x=rmvd_coll_x,
                names_x=[nm for nm, b in zip(names_after_cst, bool_coll_x) if not b], names_y=list(names_y))
