"""Host-side pre-processing, mirroring the reference's R checks and data preparation.

Follows R/prepare_atlasqtl.R:8-124 (``prepare_data_``, ``check_verbose_``,
``check_annealing_``) and R/utils.R:10-100, 276-343 (``check_*_``,
``rm_constant_``, ``rm_collinear_``).  Error messages keep the reference's
wording so tests read like the reference's own.  The argument checks run on the host;
the O(n p) work itself -- scale(X), the removal of constant and duplicated columns,
the centring of Y -- runs on the GPU (aq_prepare_data) and X stays there.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib

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


class PreparedData:
    """The standardised compact X (n x p) and the centred Y resident on the GPU (aq_prepare_data).  VbRun takes it in place
    of the X array; `Y` is the host copy of the centred responses (n x q, small) that the hyper-parameter rules need."""

    def __init__(self, handle, n, p, q, Y, device):
        self.handle, self.n, self.p, self.q, self.Y, self.device = handle, n, p, q, Y, device
        self.shape = (n, p)

    @property
    def x_ptr(self):
        return _lib.lib().aq_prep_x_device(self.handle)

    @property
    def y_ptr(self):
        return _lib.lib().aq_prep_y_device(self.handle)

    def X_host(self):
        out = np.empty((self.n, self.p), order="F")
        _lib.check(_lib.lib().aq_prep_get(self.handle, _lib.as_dp(out), None), "aq_prep_get")
        return out

    def close(self):
        if self.handle is not None:
            _lib.lib().aq_prep_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_on_device(Y, X, device=0):
    """scale(X), constant / duplicate-column removal and the centring of Y on the GPU (R/prepare_atlasqtl.R:57-83).
    X: float64 (n x p) or int8 dosages (1 byte per genotype: the fp64 matrix is then never formed on the host).
    Returns (PreparedData, bool_cst_x [p], bool_coll_x [p, original numbering], dup_of [p])."""
    import ctypes as C
    Y = np.asfortranarray(Y, dtype=np.float64)
    n, q = Y.shape
    pin = _lib.AqPrepInput()
    if np.asarray(X).dtype == np.int8:
        Xa = np.asfortranarray(X)
        pin.X, pin.X_i8 = None, Xa.ctypes.data_as(C.POINTER(C.c_int8))
    else:
        Xa = np.asfortranarray(X, dtype=np.float64)
        pin.X, pin.X_i8 = _lib.as_dp(Xa), None
    if Xa.shape[0] != n:
        raise AtlasqtlError("X and Y must have the same number of samples.")
    p = Xa.shape[1]
    pin.n, pin.p, pin.q, pin.Y, pin.device = n, p, q, _lib.as_dp(Y), int(device)
    h = C.c_void_p()
    rc = _lib.lib().aq_prepare_data(C.byref(pin), C.byref(h))
    if rc != 0:
        msg = _lib.lib().aq_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise AtlasqtlError(msg)                       # where the reference calls stop()
        raise _lib.AtlasqtlHipError(f"aq_prepare_data: [{rc}] {msg}")
    pk = C.c_int32(0)
    cst = np.zeros(p, dtype=np.uint8); coll = np.zeros(p, dtype=np.uint8); dup = np.zeros(p, dtype=np.int32)
    _lib.check(_lib.lib().aq_prep_info(h, C.byref(pk), cst.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       coll.ctypes.data_as(C.POINTER(C.c_uint8)), _lib.as_ip(dup), None, None), "aq_prep_info")
    Yc = np.empty((n, q), order="F")
    _lib.check(_lib.lib().aq_prep_get(h, None, _lib.as_dp(Yc)), "aq_prep_get")
    return PreparedData(h, n, int(pk.value), q, Yc, int(device)), cst.astype(bool), coll.astype(bool), dup


def prepare_data_(Y, X, tol, maxit, user_seed, verbose, checkpoint_path, trace_path,
                  names_x=None, names_y=None, device=0):
    """R/prepare_atlasqtl.R:8-87.  Returns dict(Y, X, bool_rmvd_x, initial_colnames_X,
    rmvd_cst_x, rmvd_coll_x, names_x, names_y); X is a PreparedData (the standardised matrix lives on the GPU), Y the
    centred responses on the host.  X may be float64 or int8 dosages."""
    check_vector_(user_seed, "user_seed", size=1, null_ok=True)
    check_vector_(tol, "tol", size=1)
    check_positive_(tol, "tol", eps=np.finfo(np.float64).eps)
    check_vector_(maxit, "maxit", size=1)
    check_natural_(maxit, "maxit")
    if not (isinstance(X, np.ndarray) and X.dtype == np.int8 and X.ndim == 2 and X.size > 0):
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

    # scale(X), rm_constant_, rm_collinear_, centring of Y: on the device (aq_prepare.hip); X stays there
    prep, bool_cst_x, bool_coll_full, dup_of = prepare_on_device(Y, X, device)
    rmvd_cst_x = [names_x[j] for j in np.where(bool_cst_x)[0]] if bool_cst_x.any() else None
    names_after_cst = [nm for nm, b in zip(names_x, bool_cst_x) if not b]
    bool_rmvd_x = bool_cst_x | bool_coll_full
    rmvd_coll_x = {names_x[j]: names_x[dup_of[j]] for j in np.where(bool_coll_full)[0]} or None   # removed name -> kept name
    return dict(Y=prep.Y, X=prep, bool_rmvd_x=bool_rmvd_x, initial_colnames_X=names_after_cst,
                rmvd_cst_x=rmvd_cst_x, rmvd_coll_x=rmvd_coll_x,
                names_x=[nm for nm, b in zip(names_x, bool_rmvd_x) if not b], names_y=list(names_y))
