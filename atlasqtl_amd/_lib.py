"""ctypes binding of libatlasqtl_hip.so (the C ABI of include/atlasqtl_hip.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing, or no HIP device is visible when a compute entry is called,
the error is raised loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AQ_LIB", os.path.join(_HERE, "libatlasqtl_hip.so"))

AQ_OK = 0
AQ_VB_DONE = 0
AQ_VB_NEED_ALLREDUCE_MAIN = 1
AQ_VB_NEED_ALLREDUCE_ELBO = 2

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


class AqVbProblem(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("p", C.c_int32), ("q", C.c_int32), ("q_total", C.c_int32),
        ("X", dp), ("Y", dp),
        ("A2_inv", C.c_double), ("m0", C.c_double), ("nu", C.c_double), ("rho", C.c_double), ("t02", C.c_double),
        ("eta", dp), ("kappa", dp), ("n0", dp),
        ("gam_vb", dp), ("mu_beta_vb", dp), ("sig02_inv_vb", C.c_double), ("sig2_beta_vb", dp),
        ("sig2_theta_vb", dp), ("tau_vb", dp), ("theta_vb", dp), ("zeta_vb", dp),
        ("has_anneal", C.c_int32), ("anneal", C.c_double * 3), ("tol", C.c_double), ("maxit", C.c_int32),
        ("thinned_elbo_eval", C.c_int32), ("debug", C.c_int32),
        ("device", C.c_int32), ("world_size", C.c_int32),
        ("ext_reduce_main", C.c_void_p), ("ext_reduce_elbo", C.c_void_p), ("init_on_device", C.c_int32),
        ("init_generate", C.c_int32), ("trait_offset", C.c_int32), ("init_seed", C.c_uint64),
        ("init_gam_mean", C.c_double), ("init_gam_sd", C.c_double),
        ("xy_on_device", C.c_int32), ("scheme", C.c_int32), ("df", C.c_int32),
    ]


class AqPrepInput(C.Structure):
    _fields_ = [("n", C.c_int32), ("p", C.c_int32), ("q", C.c_int32), ("X", dp), ("X_i8", C.POINTER(C.c_int8)), ("Y", dp),
                ("device", C.c_int32)]


class AqVbMultiOut(C.Structure):
    _fields_ = [
        ("beta_vb", dp), ("gam_vb", dp), ("mu_beta_vb", dp), ("theta_vb", dp), ("zeta_vb", dp), ("lam2_inv_vb", dp),
        ("sig2_theta_vb", dp), ("tau_vb", dp), ("sig2_beta_vb", dp), ("elbo_it", ip), ("elbo_lb", dp), ("elbo_cap", C.c_int32),
        ("n_elbo", C.c_int32), ("it", C.c_int32), ("converged", C.c_int32), ("lb_opt", C.c_double), ("diff_lb", C.c_double),
        ("sig02_inv_vb", C.c_double), ("sig2_inv_vb", C.c_double), ("seconds", C.c_double), ("core_ms", C.c_double),
    ]


class AqVbStatus(C.Structure):
    _fields_ = [
        ("it", C.c_int32), ("converged", C.c_int32), ("lb_opt", C.c_double), ("diff_lb", C.c_double),
        ("c", C.c_double), ("annealing", C.c_int32), ("n_elbo", C.c_int32), ("core_ms", C.c_double),
        ("core_launches", C.c_int32), ("sig02_inv_vb", C.c_double), ("sig2_inv_vb", C.c_double),
        ("lentz_iters", C.c_int32), ("core_kernel", C.c_int32), ("split_parts", C.c_int32),
        ("tiles_per_group", C.c_int32), ("chain_segments", C.c_int32),
    ]


# every symbol include/atlasqtl_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "aq_last_error": (C.c_char_p, []),
    "aq_version": (C.c_char_p, []),
    "aq_device_count": (C.c_int, []),
    "aq_core_dual_loop": (C.c_int, [dp, dp, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp, dp, ip, C.c_int32, ip,
                                    C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "aq_core_dual_mis_loop": (C.c_int, [dp, C.POINTER(dp), dp, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp, dp, ip,
                                        C.c_int32, ip, C.c_int32, C.c_double, C.c_int32, C.c_int32]),
    "aq_vb_reduce_len": (C.c_int64, [C.c_int32]),
    "aq_vb_create": (C.c_int, [C.POINTER(AqVbProblem), C.POINTER(C.c_void_p)]),
    "aq_vb_destroy": (None, [C.c_void_p]),
    "aq_vb_advance": (C.c_int, [C.c_void_p]),
    "aq_vb_reduce_ptr": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "aq_vb_set_sweep_budget": (C.c_int, [C.c_void_p, C.c_int32]),
    "aq_vb_run": (C.c_int, [C.c_void_p]),
    "aq_vb_run_sweeps": (C.c_int, [C.c_void_p, C.c_int32]),
    "aq_vb_get_status": (C.c_int, [C.c_void_p, C.POINTER(AqVbStatus)]),
    "aq_vb_get_overrides": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32]),
    "aq_vb_get_elbo_trace": (C.c_int32, [C.c_void_p, ip, dp, C.c_int32]),
    "aq_vb_get_result": (C.c_int, [C.c_void_p, dp, dp, dp, dp, dp, dp, dp, dp, dp]),
    "aq_vb_run_multi": (C.c_int, [C.POINTER(AqVbProblem), C.c_int32, ip, C.c_int32, C.POINTER(AqVbMultiOut)]),
    "aq_vb_partition": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, ip, ip]),
    "aq_vb_get_residual": (C.c_int, [C.c_void_p, dp]),
    "aq_prepare_data": (C.c_int, [C.POINTER(AqPrepInput), C.POINTER(C.c_void_p)]),
    "aq_prep_info": (C.c_int, [C.c_void_p, ip, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), ip, dp, dp]),
    "aq_prep_x_device": (C.c_void_p, [C.c_void_p]),
    "aq_prep_y_device": (C.c_void_p, [C.c_void_p]),
    "aq_prep_get": (C.c_int, [C.c_void_p, dp, dp]),
    "aq_prep_destroy": (None, [C.c_void_p]),
    "aq_assign_bfdr": (C.c_int, [dp, dp, C.c_int64, C.c_int32]),
    "aq_hotspot_sizes": (C.c_int, [dp, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.c_int32]),
    "aq_vb_hotspot_sizes": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "aq_vb_bfdr_begin": (C.c_int, [C.c_void_p]),
    "aq_vb_bfdr_query": (C.c_int, [C.c_void_p, C.c_double, dp]),
    "aq_vb_bfdr_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "aq_vb_bfdr_end": (None, [C.c_void_p]),
    "aq_vb_state_bytes": (C.c_int64, [C.c_void_p]),
    "aq_vb_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "aq_vb_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "aq_special_eval": (C.c_int, [C.c_int32, dp, dp, dp, C.c_int64]),
    "aq_special_eval_device": (C.c_int, [C.c_int32, dp, dp, dp, C.c_int64, C.c_int32]),
    "aq_q_approx_vec": (C.c_int, [dp, dp, C.c_int64, ip]),
    "aq_vb_debug_raise_errflag": (C.c_int, [C.c_void_p]),
}

_lib = None


class AtlasqtlHipError(RuntimeError):
    pass


def lib():
    """Load libatlasqtl_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AtlasqtlHipError(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; "
                                   "g.build()' or make -C atlasqtl_amd/csrc). There is no CPU fallback.")
        # PyTorch (used for device tensors and torch.distributed) ships its own copy of the HIP runtime.  If it is loaded AFTER this
        # library has initialised the system copy, torch.cuda no longer finds a device ("No HIP GPUs are available"); loaded first,
        # both live together.  So: torch first, when it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != AQ_OK:
        msg = lib().aq_last_error().decode("utf-8", "replace")
        raise AtlasqtlHipError(f"{what}: [{rc}] {msg}" if what else f"[{rc}] {msg}")


def as_dp(a):
    return a.ctypes.data_as(dp)


def as_ip(a):
    return a.ctypes.data_as(ip)
