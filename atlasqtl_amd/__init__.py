"""atlasqtl_amd -- MI355X-native variational-inference hot path of atlasqtl.

Public surface mirrors the reference package's exports for this path
(NAMESPACE:3-10): atlasqtl, set_hyper, set_init; plus the operator-level
coreDualLoop / coreDualMisLoop (R/RcppExports.R) backed by libatlasqtl_hip.so.
"""
from .api import atlasqtl  # noqa: F401
from .core import (VbRun, assign_bFDR, atlasqtl_global_core_, atlasqtl_global_local_core_, coreDualLoop,  # noqa: F401
                   coreDualMisLoop, hotspot_sizes)
from .hyper_init import set_hyper, set_init  # noqa: F401
from .prepare import AtlasqtlError  # noqa: F401

__all__ = ["atlasqtl", "set_hyper", "set_init", "coreDualLoop", "coreDualMisLoop", "atlasqtl_global_local_core_", "atlasqtl_global_core_",
           "VbRun", "AtlasqtlError", "assign_bFDR", "hotspot_sizes"]
