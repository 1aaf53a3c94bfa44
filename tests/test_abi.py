"""The C ABI library: loads without a GPU, exports every symbol include/atlasqtl_hip.h declares,
and its compute entries fail LOUDLY (no CPU fallback) when no HIP device is visible."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from atlasqtl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "atlasqtl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aq_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 15
    assert set(names) == set(_lib.SYMBOLS), (set(names) ^ set(_lib.SYMBOLS))


def test_library_exports_every_declared_symbol(hiplib):
    for name in declared_functions():
        assert hasattr(hiplib, name), name
    assert hiplib.aq_version().startswith(b"atlasqtl_hip")
    assert hiplib.aq_vb_reduce_len(50000) == 50000 + 8
    assert hiplib.aq_vb_reduce_len(75) == 80 + 8


def test_no_python_or_cpu_fallback_in_product():
    """The product package must not import the oracle (a product path through the oracle voids parity)."""
    pkg = os.path.join(ROOT, "atlasqtl_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, fn)).read(), fn
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        path = os.path.join(pkg, "csrc", fn)
        if os.path.isfile(path):     # sources only: _build/ holds objects
            assert "oracle" not in open(path).read().lower(), fn


def test_argument_errors_need_no_gpu(hiplib):
    assert hiplib.aq_vb_create(None, None) == 1
    assert b"NULL" in hiplib.aq_last_error()
    assert hiplib.aq_special_eval(99, _lib.as_dp(np.zeros(1)), None, _lib.as_dp(np.zeros(1)), 1) == 1


def test_compute_fails_loudly_without_device(hiplib):
    if hiplib.aq_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    import atlasqtl_amd as A
    from tests.util import make_problem, operator_inputs
    prob = make_problem(40, 12, 3, p_act=2, prob_assoc=1.0)
    with pytest.raises(_lib.AtlasqtlHipError, match="no HIP device"):
        A.atlasqtl_global_local_core_(prob["Y"], prob["X"], 3, None, 1, 0.1, 5, 0, prob["list_hyper"], prob["list_init"])
    a = operator_inputs(6, 2)
    with pytest.raises(_lib.AtlasqtlHipError, match="no HIP device"):
        A.coreDualLoop(a["cp_X"], a["cp_Y_X"], a["gam_vb"], a["log_Phi"], a["log_1mPhi"], a["log_sig2_inv_vb"],
                       a["log_tau_vb"], a["m1_beta"], a["cp_betaX_X"], a["mu_beta_vb"], a["sig2_beta_vb"],
                       a["tau_vb"], a["shuffled_ind"], a["sample_q"])
    with pytest.raises(_lib.AtlasqtlHipError):
        A.atlasqtl(prob["truth"]["Y"], prob["truth"]["X"], p0=(2, 4), verbose=0)
