"""CPU restatement (NumPy) of the O(n p) half of the reference's prepare_data_ -- test infrastructure, the checker of the
device-side aq_prepare_data (atlasqtl_amd/csrc/aq_prepare.hip); the product never imports it.

Follows R/prepare_atlasqtl.R:57-83 (scale(X), the removals, centring of Y) and R/utils.R:276-343 (rm_constant_, rm_collinear_).
Parity unpinned with respect to reference-produced numbers (R is absent from the image; the reference's tests hold no
numeric fixtures): pinned by the statement-by-statement restatement and by the invariants tested in tests/test_host_logic.py
(diag(X'X) = n - 1, column means 0, R's constant / duplicate semantics incl. non-dyadic constants).
"""
from __future__ import annotations

import numpy as np


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


def prepare_xy(Y, X):
    """R/prepare_atlasqtl.R:57-83 on arrays: returns (Xs, Yc, bool_cst_x, bool_coll_x (among the non-constant columns))."""
    X = np.asarray(X, dtype=np.float64)
    names = list(range(X.shape[1]))
    Xs = scale_columns(X)
    Xs, bool_cst_x, _ = rm_constant_(Xs, names)
    Xs, bool_coll_x, _ = rm_collinear_(Xs, [nm for nm, b in zip(names, bool_cst_x) if not b])
    Yc = np.asarray(Y, dtype=np.float64) - np.nanmean(Y, axis=0)
    return Xs, Yc, bool_cst_x, bool_coll_x
